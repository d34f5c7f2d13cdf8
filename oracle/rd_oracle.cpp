// rd_oracle.cpp -- CPU restatement of RobotDynamics.jl's batched dynamics / Jacobian hot path.
//
// THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may load it.  The product library (librdhip.so) never
// links, includes or calls anything in oracle/.
//
// What it is: a plain, per-knot-point, double-precision restatement of the reference
// algorithm, written the way the reference is written (generic `dynamics` evaluated on
// forward-mode dual numbers carrying all n+m partials at once, exactly like
// ForwardDiff.jacobian of `evaluate(fun, StaticKnotPoint(z, _x))`,
// reference src/jacobian_gen.jl:485-507), plus the literal chain-rule formulas of
// src/integration.jl:302-337.  Rotation arithmetic is written in *matrix* form
// (lmult/rmult/hmat of SURVEY App. A) so that it shares no code shape with the HIP kernels.
//
// Parity status ("pinning"): Julia is not installed in the build image and the reference
// ships no stored numeric golden vectors (its tests are property tests on rand inputs), so
// there are NO reference-generated outputs to check against.  The oracle is pinned by every
// known-answer fact the reference's own tests hold for this path (tests/test_oracle_kat.py):
//   * dimension / index tables      test/liestate.jl:33-67, test/rigidbody_test.jl:60,88-98,
//                                   test/liemodel.jl:24-26, test/rigid_body_jacobians.jl:5
//   * double-integrator Jacobians   test/double_integrator.jl:119-127,
//                                   test/discrete_dynamics_test.jl:52-63
//   * Euler/RK3/RK4 step formulas   test/integration_tests.jl:38,49-59,81-99
//   * chain rule == AD to 1e-10, finite difference within 1e-6
//                                   test/integration_tests.jl:7-18
//   * Newton-Euler components       test/rigidbody_test.jl:135-162
//   * G = blkdiag(I, L(q)H, I), nablaG = blkdiag(0, -(q.b) I, 0)
//                                   test/rigid_body_jacobians.jl:72-83, test/liestate.jl:74-99
//   * Cayley error == Rodrigues parameters of q2^-1 q1      test/rbstate.jl:157
// and cross-checked against an independent torch-autograd (fp64) restatement whose outputs
// are committed as tests/golden/*.npz ("restatement-derived", NOT "reference-derived").
// Third-party arithmetic (Rotations.jl 1.x, ForwardDiff 0.10, FiniteDiff 2.x, StaticArrays)
// is not vendored in /root/reference; its published formulas are restated here (SURVEY
// App. A).  For those formulas value parity is therefore "unpinned beyond the KATs above".
//
// Build:  g++ -O2 -fPIC -shared -fopenmp -ffp-contract=off oracle/rd_oracle.cpp -o oracle/_build/librd_oracle.so
#include <cmath>
#include <cstring>
#include <cstdio>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

// ------------------------------------------------------------------------------------------
// enums (kept numerically identical to include/rdhip.h so tests can share constants)
// ------------------------------------------------------------------------------------------
enum { M_CARTPOLE = 0, M_DOUBLE_INTEGRATOR = 1, M_LINEAR = 2, M_SATELLITE = 3, M_QUADROTOR = 4,
       M_SATELLITE_PAIR = 5, M_LIE_SATELLITE = 6,
       // explicitly time-dependent TEST models (not in the reference, whose example models are all time-invariant): they exist to check
       // that the time argument reaches every stage the way src/integration.jl passes it (t, t + h/2, t + h; implicit midpoint: residual
       // at t + h/2, Jacobian at t).  The product evaluates them as user-defined (hipRTC) models.
       M_TV_PENDULUM = 8,      // x = [th, thd]: thdd = -(a + b cos(w t)) sin(th) - c thd + u + b sin(w t)          params {a, b, w, c}
       M_TV_SATELLITE = 9,   // RigidBody satellite + disturbance: F += fd [cos wt, 0, sin wt] (world), tau += td [sin wt, cos wt, sin 2wt]   params {mass, J[9], fd, td, w}
       M_LIE_STATE = 10 };     // a bare LieState{R,P} (src/liestate.jl:75-83; test/liestate.jl:33-68): no dynamics, the error-state maps only   params {nP, P...}
enum { ROT_QUAT = 0, ROT_MRP = 1, ROT_RP = 2 };
enum { FRAME_WORLD = 0, FRAME_BODY = 1 };
enum { RULE_EULER = 0, RULE_RK2 = 1, RULE_RK3 = 2, RULE_RK4 = 3 };
enum { DIFF_FORWARD_AD = 0, DIFF_FINITE_DIFF = 1, DIFF_USER = 2 };
enum { SIG_STATIC_RETURN = 0, SIG_INPLACE = 1 };

static const int MAXN = 16;   // max state dim
static const int MAXM = 8;    // max control dim
static const int MAXZ = MAXN + MAXM;

struct OModel {
    int kind, rot, frame;
    int n, m;
    double p[64];
    int np;
};

// ------------------------------------------------------------------------------------------
// forward-mode dual number carrying NP partials (ForwardDiff.Dual{Tag,Float64,NP})
// ------------------------------------------------------------------------------------------
template <int NP>
struct Dual {
    double v;
    double d[NP];
    Dual() : v(0) { for (int i = 0; i < NP; ++i) d[i] = 0; }
    Dual(double x) : v(x) { for (int i = 0; i < NP; ++i) d[i] = 0; }
};
template <int NP> static inline Dual<NP> operator+(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r; r.v = a.v + b.v; for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int NP> static inline Dual<NP> operator-(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r; r.v = a.v - b.v; for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int NP> static inline Dual<NP> operator-(const Dual<NP>& a) {
    Dual<NP> r; r.v = -a.v; for (int i = 0; i < NP; ++i) r.d[i] = -a.d[i]; return r; }
template <int NP> static inline Dual<NP> operator*(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r; r.v = a.v * b.v; for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int NP> static inline Dual<NP> operator/(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r; r.v = a.v / b.v; double ib = 1.0 / b.v;
    for (int i = 0; i < NP; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int NP> static inline Dual<NP> operator+(const Dual<NP>& a, double b) { Dual<NP> r = a; r.v += b; return r; }
template <int NP> static inline Dual<NP> operator+(double b, const Dual<NP>& a) { Dual<NP> r = a; r.v += b; return r; }
template <int NP> static inline Dual<NP> operator-(const Dual<NP>& a, double b) { Dual<NP> r = a; r.v -= b; return r; }
template <int NP> static inline Dual<NP> operator-(double b, const Dual<NP>& a) { Dual<NP> r = -a; r.v += b; return r; }
template <int NP> static inline Dual<NP> operator*(const Dual<NP>& a, double b) {
    Dual<NP> r; r.v = a.v * b; for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] * b; return r; }
template <int NP> static inline Dual<NP> operator*(double b, const Dual<NP>& a) { return a * b; }
template <int NP> static inline Dual<NP> operator/(const Dual<NP>& a, double b) {
    Dual<NP> r; r.v = a.v / b; for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] / b; return r; }
template <int NP> static inline Dual<NP> operator/(double a, const Dual<NP>& b) { return Dual<NP>(a) / b; }
template <int NP> static inline Dual<NP> sin_(const Dual<NP>& a) {
    Dual<NP> r; r.v = std::sin(a.v); double c = std::cos(a.v); for (int i = 0; i < NP; ++i) r.d[i] = c * a.d[i]; return r; }
template <int NP> static inline Dual<NP> cos_(const Dual<NP>& a) {
    Dual<NP> r; r.v = std::cos(a.v); double s = -std::sin(a.v); for (int i = 0; i < NP; ++i) r.d[i] = s * a.d[i]; return r; }
template <int NP> static inline Dual<NP> sqrt_(const Dual<NP>& a) {
    Dual<NP> r; r.v = std::sqrt(a.v); double c = 0.5 / r.v; for (int i = 0; i < NP; ++i) r.d[i] = c * a.d[i]; return r; }
// ForwardDiff: max(x::Real, d::Dual) == ifelse(x > d, x, d)   
template <int NP> static inline Dual<NP> max0_(const Dual<NP>& a) { return (0.0 > a.v) ? Dual<NP>(0.0) : a; }
static inline double sin_(double a) { return std::sin(a); }
static inline double cos_(double a) { return std::cos(a); }
static inline double sqrt_(double a) { return std::sqrt(a); }
static inline double max0_(double a) { return (0.0 > a) ? 0.0 : a; }
static inline double val(double a) { return a; }
template <int NP> static inline double val(const Dual<NP>& a) { return a.v; }

// ------------------------------------------------------------------------------------------
// Rotations.jl primitives in matrix form (scalar-first quaternion [w,x,y,z])
// ------------------------------------------------------------------------------------------
template <class S> static void lmult(const S q[4], S L[4][4]) {   // L(q) p = q (x) p
    const S &w = q[0], &x = q[1], &y = q[2], &z = q[3];
    S M[4][4] = {{w, -x, -y, -z}, {x, w, -z, y}, {y, z, w, -x}, {z, -y, x, w}};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) L[i][j] = M[i][j];
}
template <class S> static void rmult(const S q[4], S R[4][4]) {   // R(q) p = p (x) q
    const S &w = q[0], &x = q[1], &y = q[2], &z = q[3];
    S M[4][4] = {{w, -x, -y, -z}, {x, w, z, -y}, {y, -z, w, x}, {z, y, -x, w}};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) R[i][j] = M[i][j];
}
// q1 (x) q2 = L(q1) q2
template <class S> static void qmul(const S a[4], const S b[4], S out[4]) {
    S L[4][4]; lmult(a, L);
    for (int i = 0; i < 4; ++i) { S acc = L[i][0] * b[0]; for (int j = 1; j < 4; ++j) acc = acc + L[i][j] * b[j]; out[i] = acc; }
}
// Rotations.kinematics(q, w) = 1/2 * L(q) * [0; w]   (src/rigidbody.jl:224)
template <class S> static void kinematics(const S q[4], const S w[3], S qd[4]) {
    S p[4] = {S(0.0), w[0], w[1], w[2]}; S t[4]; qmul(q, p, t);
    for (int i = 0; i < 4; ++i) qd[i] = t[i] * 0.5;
}
template <class S> static void cross3(const S a[3], const S b[3], S c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
// q * r for a (possibly NON-unit) QuatRotation:  (w^2 - v.v) r + 2 v (v.r) + 2 w (v x r)
template <class S> static void qrot(const S q[4], const S r[3], S out[3]) {
    const S& w = q[0]; const S* v = q + 1;
    S vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    S vr = v[0] * r[0] + v[1] * r[1] + v[2] * r[2];
    S c[3]; cross3(v, r, c);
    S a = w * w - vv;
    for (int i = 0; i < 3; ++i) out[i] = a * r[i] + 2.0 * (v[i] * vr) + 2.0 * (w * c[i]);
}
// q \ r = conj(q) * r   (inv of a QuatRotation is the conjugate, no normalisation)
template <class S> static void qrot_inv(const S q[4], const S r[3], S out[3]) {
    S qc[4] = {q[0], -q[1], -q[2], -q[3]}; qrot(qc, r, out);
}

template <class S> static void matvec3(const double A[9], const S x[3], S y[3]) {   // A column-major
    for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[i + 3] * x[1] + A[i + 6] * x[2];
}
static bool is_diag3(const double A[9]) { return A[1] == 0 && A[2] == 0 && A[3] == 0 && A[5] == 0 && A[6] == 0 && A[7] == 0; }
static void inv3(const double A[9], double Ai[9]) {
    if (is_diag3(A)) {   // inv(Diagonal) = Diagonal(1 ./ d)
        for (int i = 0; i < 9; ++i) Ai[i] = 0; Ai[0] = 1.0 / A[0]; Ai[4] = 1.0 / A[4]; Ai[8] = 1.0 / A[8]; return;
    }
    double a = A[0], b = A[3], c = A[6], d = A[1], e = A[4], f = A[7], g = A[2], h = A[5], i = A[8];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    Ai[0] = (e * i - f * h) / det; Ai[3] = (c * h - b * i) / det; Ai[6] = (b * f - c * e) / det;
    Ai[1] = (f * g - d * i) / det; Ai[4] = (a * i - c * g) / det; Ai[7] = (c * d - a * f) / det;
    Ai[2] = (d * h - e * g) / det; Ai[5] = (b * g - a * h) / det; Ai[8] = (a * e - b * d) / det;
}

// ------------------------------------------------------------------------------------------
// continuous dynamics  xdot = f(x,u,t)
// ------------------------------------------------------------------------------------------
// Cartpole: examples/cartpole.jl:15-34 (== test/cartpole_model.jl:11-30); params (mc,mp,l,g)
template <class S> static void dyn_cartpole(const OModel& M, const S* x, const S* u, S* xd) {
    double mc = M.p[0], mp = M.p[1], l = M.p[2], g = M.p[3];
    S qd0 = x[2], qd1 = x[3];
    S s = sin_(x[1]), c = cos_(x[1]);
    S H11 = S(mc + mp), H12 = (mp * l) * c, H21 = (mp * l) * c, H22 = S(mp * l * l);
    S C12 = (-mp) * qd1 * l * s;
    S r0 = C12 * qd1 + 0.0 - u[0];            // C*qd + G - B*u
    S r1 = (mp * g * l) * s;
    // StaticArrays 2x2 solve: d = det(H); [ (H22 b1 - H12 b2)/d , (H11 b2 - H21 b1)/d ]
    S det = H11 * H22 - H12 * H21;
    S y0 = (H22 * r0 - H12 * r1) / det;
    S y1 = (H11 * r1 - H21 * r0) / det;
    xd[0] = qd0; xd[1] = qd1; xd[2] = -y0; xd[3] = -y1;
}
// Double integrator DI{D}: test/double_integrator.jl:97-107; params (D)
template <class S> static void dyn_di(const OModel& M, const S* x, const S* u, S* xd) {
    int D = M.m;
    for (int i = 0; i < D; ++i) xd[i] = x[D + i];
    for (int i = 0; i < D; ++i) xd[D + i] = u[i];
}
// Linear xdot = A x + B u (new KAT model); params (n,m,A col-major,B col-major)
template <class S> static void dyn_linear(const OModel& M, const S* x, const S* u, S* xd) {
    int n = M.n, m = M.m; const double* A = M.p + 2; const double* B = A + n * n;
    for (int i = 0; i < n; ++i) {
        S acc = S(0.0);
        for (int j = 0; j < n; ++j) acc = acc + A[i + j * n] * x[j];
        for (int j = 0; j < m; ++j) acc = acc + B[i + j * n] * u[j];
        xd[i] = acc;
    }
}
// wrenches: Satellite examples/single_satellite.jl:18-27 ; Quadrotor test/quadrotor.jl:56-96
template <class S> static void wrench_satellite(const OModel&, const S* x, const S* u, S F[3], S T[3]) {
    S f[3] = {u[0], u[1], u[2]}; qrot(x + 3, f, F);
    T[0] = u[3]; T[1] = u[4]; T[2] = u[5];
}
template <class S> static void wrench_tv_satellite(const OModel& M, const S* x, const S* u, S t, S F[3], S T[3]) {
    wrench_satellite(M, x, u, F, T);
    const double fd = M.p[10], td = M.p[11], w = M.p[12];
    F[0] = F[0] + fd * cos_(w * t); F[2] = F[2] + fd * sin_(w * t);
    T[0] = T[0] + td * sin_(w * t); T[1] = T[1] + td * cos_(w * t); T[2] = T[2] + td * sin_((2.0 * w) * t);
}
template <class S> static void dyn_tv_pendulum(const OModel& M, const S* x, const S* u, S t, S* xd) {
    const double a = M.p[0], b = M.p[1], w = M.p[2], c = M.p[3];
    xd[0] = x[1];
    xd[1] = -((a + b * cos_(w * t)) * sin_(x[0])) - c * x[1] + u[0] + b * sin_(w * t);
}
template <class S> static void wrench_quadrotor(const OModel& M, const S* x, const S* u, S F[3], S T[3]) {
    double mass = M.p[0]; const double* g = M.p + 10; double L = M.p[13], kf = M.p[14], km = M.p[15];
    S F1 = max0_(kf * u[0]), F2 = max0_(kf * u[1]), F3 = max0_(kf * u[2]), F4 = max0_(kf * u[3]);
    S Fb[3] = {S(0.0), S(0.0), F1 + F2 + F3 + F4}; S qF[3]; qrot(x + 3, Fb, qF);
    for (int i = 0; i < 3; ++i) F[i] = mass * g[i] + qF[i];
    S M1 = km * u[0], M2 = km * u[1], M3 = km * u[2], M4 = km * u[3];
    T[0] = L * (F2 - F4); T[1] = L * (F3 - F1); T[2] = (M1 - M2 + M3 - M4);
}
// RigidBody{QuatRotation} Newton-Euler: src/rigidbody.jl:213-236 ; params (mass, J[9] col-major, ...)
template <class S> static void dyn_rigidbody(const OModel& M, const S* x, const S* u, S t, S* xd) {
    double mass = M.p[0]; const double* J = M.p + 1; double Jinv[9]; inv3(J, Jinv);
    const S* q = x + 3; const S* v = x + 7; const S* w = x + 10;
    S F[3], T[3];
    if (M.kind == M_SATELLITE) wrench_satellite(M, x, u, F, T);
    else if (M.kind == M_TV_SATELLITE) wrench_tv_satellite(M, x, u, t, F, T);
    else wrench_quadrotor(M, x, u, F, T);
    S qd[4]; kinematics(q, w, qd);
    S rd[3], vd[3];
    if (M.frame == FRAME_WORLD) {
        for (int i = 0; i < 3; ++i) { rd[i] = v[i]; vd[i] = F[i] / mass; }
    } else {
        qrot(q, v, rd);
        S Fm[3] = {F[0] / mass, F[1] / mass, F[2] / mass}; S a[3]; qrot_inv(q, Fm, a);
        S c[3]; cross3(w, v, c);
        for (int i = 0; i < 3; ++i) vd[i] = a[i] - c[i];
    }
    S Jw[3]; matvec3(J, w, Jw); S c[3]; cross3(w, Jw, c);
    S tt[3] = {T[0] - c[0], T[1] - c[1], T[2] - c[2]}; S wd[3]; matvec3(Jinv, tt, wd);
    for (int i = 0; i < 3; ++i) { xd[i] = rd[i]; xd[7 + i] = vd[i]; xd[10 + i] = wd[i]; }
    for (int i = 0; i < 4; ++i) xd[3 + i] = qd[i];
}
// RigidBody{MRP} / RigidBody{RodriguesParam}: the same Newton-Euler equations (src/rigidbody.jl:213-236) on x = [r(3) p(3) v(3) w(3)]
// (gen_inds src/rigidbody.jl:80-94).  Rotations.jl [3P, SURVEY App. A]: a three-parameter rotation acts on a vector through its unit
// quaternion -- MRP: q = [1 - p.p, 2p] / (1 + p.p); RodriguesParam: q = [1, g] / sqrt(1 + g.g) -- and its kinematics are
// pdot = 1/4 A(p) w with A = (1 - p.p) I + 2 (skew(p) + p p'), gdot = 1/2 (I + skew(g) + g g') w; A and I + skew + gg' are also
// grad-differential of the two types (oracle_errstate_jacobian).  Written with the explicit 3 x 3 matrices.
template <class S> static void rot3_quat(int rot, const S p[3], S q[4]) {
    S n = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if (rot == ROT_MRP) { S d = 1.0 + n; q[0] = (1.0 - n) / d; for (int i = 0; i < 3; ++i) q[1 + i] = 2.0 * p[i] / d; }
    else { S d = sqrt_(1.0 + n); q[0] = S(1.0) / d; for (int i = 0; i < 3; ++i) q[1 + i] = p[i] / d; }
}
template <class S> static void rot3_gradient(int rot, const S p[3], S A[3][3]) {    // grad-differential(p)
    S n = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    S sk[3][3] = {{S(0.0), -p[2], p[1]}, {p[2], S(0.0), -p[0]}, {-p[1], p[0], S(0.0)}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        if (rot == ROT_MRP) A[i][j] = ((i == j) ? 1.0 - n : S(0.0)) + 2.0 * (sk[i][j] + p[i] * p[j]);
        else A[i][j] = ((i == j) ? S(1.0) : S(0.0)) + sk[i][j] + p[i] * p[j];
    }
}
template <class S> static void dyn_rigidbody3(const OModel& M, const S* x, const S* u, S* xd) {
    double mass = M.p[0]; const double* J = M.p + 1; double Jinv[9]; inv3(J, Jinv);
    const S* p = x + 3; const S* v = x + 6; const S* w = x + 9;
    S q[4]; rot3_quat(M.rot, p, q);
    S F[3], T[3];
    if (M.kind == M_SATELLITE) { S f[3] = {u[0], u[1], u[2]}; qrot(q, f, F); T[0] = u[3]; T[1] = u[4]; T[2] = u[5]; }
    else {
        const double* g = M.p + 10; double L = M.p[13], kf = M.p[14], km = M.p[15];
        S F1 = max0_(kf * u[0]), F2 = max0_(kf * u[1]), F3 = max0_(kf * u[2]), F4 = max0_(kf * u[3]);
        S Fb[3] = {S(0.0), S(0.0), F1 + F2 + F3 + F4}; S qF[3]; qrot(q, Fb, qF);
        for (int i = 0; i < 3; ++i) F[i] = mass * g[i] + qF[i];
        T[0] = L * (F2 - F4); T[1] = L * (F3 - F1); T[2] = km * u[0] - km * u[1] + km * u[2] - km * u[3];
    }
    S A[3][3]; rot3_gradient(M.rot, p, A);
    const double kfac = M.rot == ROT_MRP ? 0.25 : 0.5;
    for (int i = 0; i < 3; ++i) xd[3 + i] = kfac * (A[i][0] * w[0] + A[i][1] * w[1] + A[i][2] * w[2]);
    if (M.frame == FRAME_WORLD) {
        for (int i = 0; i < 3; ++i) { xd[i] = v[i]; xd[6 + i] = F[i] / mass; }
    } else {
        S rd[3]; qrot(q, v, rd);
        S Fm[3] = {F[0] / mass, F[1] / mass, F[2] / mass}; S a[3]; qrot_inv(q, Fm, a);
        S c[3]; cross3(w, v, c);
        for (int i = 0; i < 3; ++i) { xd[i] = rd[i]; xd[6 + i] = a[i] - c[i]; }
    }
    S Jw[3]; matvec3(J, w, Jw); S c[3]; cross3(w, Jw, c);
    S tt[3] = {T[0] - c[0], T[1] - c[1], T[2] - c[2]}; S wd[3]; matvec3(Jinv, tt, wd);
    for (int i = 0; i < 3; ++i) xd[9 + i] = wd[i];
}
// SatellitePair LieState(R,(0,3,3)): examples/satellite_pair.jl:14-38; params (J1[9], J2[9])
template <class S> static void dyn_satpair(const OModel& M, const S* x, const S* u, S* xd) {
    for (int b = 0; b < 2; ++b) {
        const double* J = M.p + 9 * b; double Jinv[9]; inv3(J, Jinv);
        const S* q = x + 7 * b; const S* w = x + 7 * b + 4; const S* ub = u + 3 * b;
        S Jw[3]; matvec3(J, w, Jw); S c[3]; cross3(w, Jw, c);
        S tt[3] = {ub[0] - c[0], ub[1] - c[1], ub[2] - c[2]}; S wd[3]; matvec3(Jinv, tt, wd);
        S qd[4]; kinematics(q, w, qd);
        for (int i = 0; i < 4; ++i) xd[7 * b + i] = qd[i];
        for (int i = 0; i < 3; ++i) xd[7 * b + 4 + i] = wd[i];
    }
}
// Lie "Satellite" of test/liemodel.jl:1-22: x=[w;q], LieState(R,(3,0)), q normalised inside; params (Jdiag[3])
template <class S> static void dyn_liesat(const OModel& M, const S* x, const S* u, S* xd) {
    const double* Jd = M.p; const S* w = x;
    S nn = sqrt_(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
    S q[4] = {x[3] / nn, x[4] / nn, x[5] / nn, x[6] / nn};
    S Jw[3] = {Jd[0] * w[0], Jd[1] * w[1], Jd[2] * w[2]}; S c[3]; cross3(w, Jw, c);
    for (int i = 0; i < 3; ++i) xd[i] = (u[i] - c[i]) / Jd[i];
    S qd[4]; kinematics(q, w, qd);   // 0.5 * lmult(q) * hmat() * w
    for (int i = 0; i < 4; ++i) xd[3 + i] = qd[i];
}

// dynamics(model, x, u, t): the reference's models drop t (src/dynamics.jl:83); the two test models use it
template <class S> static void dynamics(const OModel& M, const S* x, const S* u, S t, S* xd) {
    switch (M.kind) {
        case M_CARTPOLE: dyn_cartpole(M, x, u, xd); break;
        case M_DOUBLE_INTEGRATOR: dyn_di(M, x, u, xd); break;
        case M_LINEAR: dyn_linear(M, x, u, xd); break;
        case M_SATELLITE: case M_QUADROTOR: case M_TV_SATELLITE:
            if (M.rot != ROT_QUAT && M.kind != M_TV_SATELLITE) dyn_rigidbody3(M, x, u, xd); else dyn_rigidbody(M, x, u, t, xd);
            break;
        case M_TV_PENDULUM: dyn_tv_pendulum(M, x, u, t, xd); break;
        case M_SATELLITE_PAIR: dyn_satpair(M, x, u, xd); break;
        case M_LIE_SATELLITE: dyn_liesat(M, x, u, xd); break;
    }
}

// ------------------------------------------------------------------------------------------
// explicit integrators  (src/integration.jl:73-83 Euler, :130-147 RK3, :280-300 RK4;
// RK2 = v0.3 explicit midpoint, SURVEY App. B / test/integration_tests.jl:42-45)
// sig==SIG_STATIC_RETURN follows `integrate` (k_i = f*h ; x + k1/2), sig==SIG_INPLACE follows
// `integrate!` (k_i = f ; x + k1*h/2) -- they differ only in rounding.
// ------------------------------------------------------------------------------------------
template <class S> static void integrate(const OModel& M, int rule, int sig, const S* x, const S* u, S t, S h, S* xn) {
    int n = M.n; S k1[MAXN], k2[MAXN], k3[MAXN], k4[MAXN], xs[MAXN];
    if (rule == RULE_EULER) {
        dynamics(M, x, u, t, k1);
        for (int i = 0; i < n; ++i) xn[i] = (sig == SIG_STATIC_RETURN) ? x[i] + h * k1[i] : k1[i] * h + x[i];
    } else if (rule == RULE_RK2) {
        dynamics(M, x, u, t, k1);
        for (int i = 0; i < n; ++i) xs[i] = (sig == SIG_STATIC_RETURN) ? x[i] + (k1[i] * h) / 2.0 : x[i] + k1[i] * h / 2.0;
        dynamics(M, xs, u, t + h / 2.0, k2);
        for (int i = 0; i < n; ++i) xn[i] = x[i] + k2[i] * h;
    } else if (rule == RULE_RK3) {
        if (sig == SIG_STATIC_RETURN) {
            dynamics(M, x, u, t, k1); for (int i = 0; i < n; ++i) { k1[i] = k1[i] * h; xs[i] = x[i] + k1[i] / 2.0; }
            dynamics(M, xs, u, t + h / 2.0, k2); for (int i = 0; i < n; ++i) { k2[i] = k2[i] * h; xs[i] = x[i] - k1[i] + 2.0 * k2[i]; }
            dynamics(M, xs, u, t + h, k3); for (int i = 0; i < n; ++i) k3[i] = k3[i] * h;
            for (int i = 0; i < n; ++i) xn[i] = x[i] + (k1[i] + 4.0 * k2[i] + k3[i]) / 6.0;
        } else {
            dynamics(M, x, u, t, k1); for (int i = 0; i < n; ++i) xs[i] = x[i] + k1[i] * h / 2.0;
            dynamics(M, xs, u, t + h / 2.0, k2); for (int i = 0; i < n; ++i) xs[i] = x[i] - k1[i] * h + 2.0 * k2[i] * h;
            dynamics(M, xs, u, t + h, k3);
            for (int i = 0; i < n; ++i) xn[i] = x[i] + h * (k1[i] + 4.0 * k2[i] + k3[i]) / 6.0;
        }
    } else {
        if (sig == SIG_STATIC_RETURN) {
            dynamics(M, x, u, t, k1); for (int i = 0; i < n; ++i) { k1[i] = k1[i] * h; xs[i] = x[i] + k1[i] / 2.0; }
            dynamics(M, xs, u, t + h / 2.0, k2); for (int i = 0; i < n; ++i) { k2[i] = k2[i] * h; xs[i] = x[i] + k2[i] / 2.0; }
            dynamics(M, xs, u, t + h / 2.0, k3); for (int i = 0; i < n; ++i) { k3[i] = k3[i] * h; xs[i] = x[i] + k3[i]; }
            dynamics(M, xs, u, t + h, k4); for (int i = 0; i < n; ++i) k4[i] = k4[i] * h;
            for (int i = 0; i < n; ++i) xn[i] = x[i] + (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) / 6.0;
        } else {
            dynamics(M, x, u, t, k1); for (int i = 0; i < n; ++i) xs[i] = x[i] + k1[i] * h / 2.0;
            dynamics(M, xs, u, t + h / 2.0, k2); for (int i = 0; i < n; ++i) xs[i] = x[i] + k2[i] * h / 2.0;
            dynamics(M, xs, u, t + h / 2.0, k3); for (int i = 0; i < n; ++i) xs[i] = x[i] + k3[i] * h;
            dynamics(M, xs, u, t + h, k4);
            for (int i = 0; i < n; ++i) xn[i] = x[i] + h * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) / 6.0;
        }
    }
}

// `evaluate(fun, z)`: fun continuous (rule < 0) or DiscretizedDynamics{rule}
// terminal knot point (dt === 0): control reads as zeros (src/knotpoint.jl:67,123)
template <class S> static void evaluate(const OModel& M, int rule, int sig, const S* z, double t, double dt, S* y) {
    S u[MAXM]; const S* x = z;
    for (int i = 0; i < M.m; ++i) u[i] = (dt == 0.0) ? S(0.0) : z[M.n + i];
    if (rule < 0) dynamics(M, x, u, S(t), y);
    else integrate(M, rule, sig, x, u, S(t), S(dt), y);
}

// ------------------------------------------------------------------------------------------
// analytic continuous Jacobians (DiffMethod UserDefined)
// ------------------------------------------------------------------------------------------
// J is n x (n+m) column-major (src/jacobian.jl:26-37)
#define JJ(i, j) J[(i) + (j) * n]
static void skew3(const double v[3], double S[3][3]) {
    S[0][0] = 0; S[0][1] = -v[2]; S[0][2] = v[1]; S[1][0] = v[2]; S[1][1] = 0; S[1][2] = -v[0]; S[2][0] = -v[1]; S[2][1] = v[0]; S[2][2] = 0;
}
// Rotations.grad-rotate(q, r) = 2 V R(q)^T R([0;r])  (3x4)   (test/rigidbody_test.jl:12-17)
static void grad_rotate(const double q[4], const double r[3], double G[3][4]) {
    double Rq[4][4], Rr[4][4]; double rh[4] = {0, r[0], r[1], r[2]}; rmult(q, Rq); rmult(rh, Rr);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) {
        double acc = 0; for (int k = 0; k < 4; ++k) acc += Rq[k][i + 1] * Rr[k][j]; G[i][j] = 2 * acc; }
}
static void rotmat(const double q[4], double R[3][3]) {   // columns = q * e_j
    for (int j = 0; j < 3; ++j) { double e[3] = {0, 0, 0}; e[j] = 1; double c[3]; qrot(q, e, c); for (int i = 0; i < 3; ++i) R[i][j] = c[i]; }
}

static int user_jacobian(const OModel& M, const double* x, const double* u, double t, double* J) {
    int n = M.n, m = M.m; for (int i = 0; i < n * (n + m); ++i) J[i] = 0;
    if (M.kind == M_TV_PENDULUM) {   // A(t) = [0 1; -(a + b cos wt) cos th, -c], B = [0; 1]
        JJ(0, 1) = 1.0; JJ(1, 0) = -((M.p[0] + M.p[1] * std::cos(M.p[2] * t)) * std::cos(x[0])); JJ(1, 1) = -M.p[3]; JJ(1, 2) = 1.0;
        return 0;
    }
    if (M.kind == M_CARTPOLE) {   // test/cartpole_model.jl:57-96
        double mc = M.p[0], mp = M.p[1], l = M.p[2], g = M.p[3];
        double s = std::sin(x[1]), c = std::cos(x[1]); double qd[2] = {x[2], x[3]};
        double H[2][2] = {{mc + mp, mp * l * c}, {mp * l * c, mp * l * l}};
        double xd[4]; dyn_cartpole(M, x, u, xd); double qdd[2] = {xd[2], xd[3]};
        double dH[2][5] = {{0, -mp * l * s * qdd[1], 0, 0, 0}, {0, -mp * l * s * qdd[0], 0, 0, 0}};
        double dC[2][5] = {{0, -mp * l * c * qd[1] * qd[1], 0, -2 * mp * l * qd[1] * s, 0}, {0, 0, 0, 0, 0}};
        double dG[2][5] = {{0, 0, 0, 0, 0}, {0, mp * g * l * c, 0, 0, 0}};
        double dB[2][5] = {{0, 0, 0, 0, 1}, {0, 0, 0, 0, 0}};
        JJ(0, 2) = 1.0; JJ(1, 3) = 1.0;
        double det = H[0][0] * H[1][1] - H[0][1] * H[1][0];
        for (int j = 0; j < 5; ++j) {
            double b0 = -dH[0][j] - dC[0][j] - dG[0][j] + dB[0][j], b1 = -dH[1][j] - dC[1][j] - dG[1][j] + dB[1][j];
            JJ(2, j) = (H[1][1] * b0 - H[0][1] * b1) / det; JJ(3, j) = (H[0][0] * b1 - H[1][0] * b0) / det;
        }
        return 0;
    }
    if (M.kind == M_DOUBLE_INTEGRATOR) {   // test/double_integrator.jl:119-127: J[i, i+M] = 1
        for (int i = 0; i < n; ++i) JJ(i, i + m) = 1.0; return 0;
    }
    if (M.kind == M_LINEAR) {
        const double* A = M.p + 2; const double* B = A + n * n;
        for (int i = 0; i < n * n; ++i) J[i] = A[i]; for (int i = 0; i < n * m; ++i) J[n * n + i] = B[i]; return 0;
    }
    if ((M.kind == M_SATELLITE || M.kind == M_QUADROTOR) && M.rot != ROT_QUAT) return -1;   // jacobian!(::RigidBody{<:QuatRotation}, ...) only (src/rigidbody.jl:260)
    if (M.kind == M_SATELLITE || M.kind == M_QUADROTOR || M.kind == M_TV_SATELLITE) {      // (the test model's disturbance is additive: same Jacobian)
        // src/rigidbody.jl:260-313 (formulas; the reference method itself references an undefined `z`
        // at :278 and cannot run -- the formulas below are its intended content with the wrench
        // Jacobian supplied analytically: examples/single_satellite.jl:49-61 for the satellite).
        double mass = M.p[0]; const double* Jm = M.p + 1; double Jinv[9]; inv3(Jm, Jinv);
        const double* q = x + 3; const double* v = x + 7; const double* w = x + 10;
        double Fw[3], Tw[3];
        double Wj[6][MAXZ]; for (int i = 0; i < 6; ++i) for (int j = 0; j < n + m; ++j) Wj[i][j] = 0;
        double Rq[3][3]; rotmat(q, Rq);
        if (M.kind == M_SATELLITE || M.kind == M_TV_SATELLITE) {
            if (M.kind == M_TV_SATELLITE) wrench_tv_satellite(M, x, u, t, Fw, Tw); else wrench_satellite(M, x, u, Fw, Tw);
            double Gr[3][4]; grad_rotate(q, u, Gr);
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 4; ++j) Wj[i][3 + j] = Gr[i][j]; for (int j = 0; j < 3; ++j) Wj[i][13 + j] = Rq[i][j]; Wj[3 + i][16 + i] = 1.0; }
        } else {
            wrench_quadrotor(M, x, u, Fw, Tw);
            double L = M.p[13], kf = M.p[14], km = M.p[15];
            double act[4], Fs = 0; for (int k = 0; k < 4; ++k) { act[k] = (0.0 > kf * u[k]) ? 0.0 : 1.0; Fs += act[k] * kf * u[k]; }
            double Fb[3] = {0, 0, Fs}; double Gr[3][4]; grad_rotate(q, Fb, Gr);
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 4; ++j) Wj[i][3 + j] = Gr[i][j]; for (int k = 0; k < 4; ++k) Wj[i][13 + k] = Rq[i][2] * kf * act[k]; }
            Wj[3][13 + 1] = L * kf * act[1]; Wj[3][13 + 3] = -L * kf * act[3];
            Wj[4][13 + 2] = L * kf * act[2]; Wj[4][13 + 0] = -L * kf * act[0];
            Wj[5][13 + 0] = km; Wj[5][13 + 1] = -km; Wj[5][13 + 2] = km; Wj[5][13 + 3] = -km;
        }
        // force rows scaled by tmp = I/m (world) or RotMatrix(inv(q))/m (body); moment rows by Jinv
        double tmp[3][3];
        if (M.frame == FRAME_WORLD) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) tmp[i][j] = (i == j) ? 1.0 / mass : 0.0; }
        else { double qc[4] = {q[0], -q[1], -q[2], -q[3]}; double Rc[3][3]; rotmat(qc, Rc); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) tmp[i][j] = Rc[i][j] / mass; }
        for (int j = 0; j < n + m; ++j) for (int i = 0; i < 3; ++i) {
            double a = 0, b = 0; for (int k = 0; k < 3; ++k) { a += tmp[i][k] * Wj[k][j]; b += Jinv[i + 3 * k] * Wj[3 + k][j]; }
            JJ(7 + i, j) = a; JJ(10 + i, j) = b;
        }
        // F[iq,iq] = 0.5 rmult(pure(w)); F[iq,iw] = 0.5 lmult(q) hmat
        double ph[4] = {0, w[0], w[1], w[2]}; double Rw[4][4], Lq[4][4]; rmult(ph, Rw); lmult(q, Lq);
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) JJ(3 + i, 3 + j) = 0.5 * Rw[i][j]; for (int j = 0; j < 3; ++j) JJ(3 + i, 10 + j) = 0.5 * Lq[i][j + 1]; }
        // F[iw,iw] += Jinv (skew(J w) - skew(w) J)
        double Jw[3]; matvec3(Jm, w, Jw); double S1[3][3], S2[3][3]; skew3(Jw, S1); skew3(w, S2);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double acc = 0; for (int k = 0; k < 3; ++k) { double s2J = 0; for (int l2 = 0; l2 < 3; ++l2) s2J += S2[k][l2] * Jm[l2 + 3 * j]; acc += Jinv[i + 3 * k] * (S1[k][j] - s2J); }
            JJ(10 + i, 10 + j) += acc;
        }
        if (M.frame == FRAME_WORLD) { JJ(0, 7) += 1; JJ(1, 8) += 1; JJ(2, 9) += 1; }
        else {
            double Gv[3][4]; grad_rotate(q, v, Gv);
            double qc[4] = {q[0], -q[1], -q[2], -q[3]}; double Gf[3][4]; grad_rotate(qc, Fw, Gf);
            double Sw[3][3], Sv[3][3]; skew3(w, Sw); skew3(v, Sv);
            const double T[4] = {1, -1, -1, -1};
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 4; ++j) { JJ(i, 3 + j) += Gv[i][j]; JJ(7 + i, 3 + j) += Gf[i][j] * T[j] / mass; }
                for (int j = 0; j < 3; ++j) { JJ(i, 7 + j) += Rq[i][j]; JJ(7 + i, 7 + j) += -Sw[i][j]; JJ(7 + i, 10 + j) += Sv[i][j]; }
            }
        }
        return 0;
    }
    return -1;   // NotImplementedError("User-defined Jacobian not implemented") src/functionbase.jl:257
}
#undef JJ

// wrenches(model, x, u, t) = [forces; moments] (src/rigidbody.jl:244-248) of the RigidBody models, and its ForwardDiff Jacobian with respect to
// z = [x; u] -- what the default wrench_jacobian!(F, model, z) of src/rigidbody.jl:315-320 is written to return (in v0.4.8 it cannot run: the
// knot-point method of `wrenches` it calls is commented out at :243); test/rigidbody_test.jl:33-46 holds the closed form for the satellite wrench.
template <class S> static int wrench_any(const OModel& M, const S* x, const S* u, S t, S* W) {
    if (M.kind != M_SATELLITE && M.kind != M_QUADROTOR && M.kind != M_TV_SATELLITE) return -1;
    S F[3], T[3];
    if (M.rot == ROT_QUAT) {
        if (M.kind == M_SATELLITE) wrench_satellite(M, x, u, F, T);
        else if (M.kind == M_TV_SATELLITE) wrench_tv_satellite(M, x, u, t, F, T);
        else wrench_quadrotor(M, x, u, F, T);
    } else {
        // three-parameter attitudes act through their unit quaternion (dyn_rigidbody3): the quaternion wrench on [r, q(p), ...]
        if (M.kind == M_TV_SATELLITE) return -1;
        S xx[7]; for (int i = 0; i < 3; ++i) xx[i] = x[i];
        rot3_quat(M.rot, x + 3, xx + 3);
        if (M.kind == M_SATELLITE) wrench_satellite(M, xx, u, F, T); else wrench_quadrotor(M, xx, u, F, T);
    }
    for (int i = 0; i < 3; ++i) { W[i] = F[i]; W[3 + i] = T[i]; }
    return 0;
}
template <int NP> static int wrench_jac_ad(const OModel& M, const double* z, double t, double* F, double* w) {
    Dual<NP> zd[MAXZ], wd[6];
    for (int i = 0; i < NP; ++i) { zd[i] = Dual<NP>(z[i]); zd[i].d[i] = 1.0; }
    int rc = wrench_any(M, zd, zd + M.n, Dual<NP>(t), wd);
    if (rc) return rc;
    for (int i = 0; i < 6; ++i) { if (w) w[i] = wd[i].v; for (int j = 0; j < NP; ++j) F[i + j * 6] = wd[i].d[j]; }
    return 0;
}
typedef int (*wrench_jac_fn)(const OModel&, const double*, double, double*, double*);
template <int NP> struct WrenchTable { static void fill(wrench_jac_fn* tab) { tab[NP] = &wrench_jac_ad<NP>; WrenchTable<NP - 1>::fill(tab); } };
template <> struct WrenchTable<1> { static void fill(wrench_jac_fn* tab) { tab[1] = &wrench_jac_ad<1>; } };

// ------------------------------------------------------------------------------------------
// Jacobians of `evaluate`
// ------------------------------------------------------------------------------------------
template <int NP> static void jac_forward_ad(const OModel& M, int rule, int sig, const double* z, double t, double dt, double* J, double* y) {
    int n = M.n; Dual<NP> zd[MAXZ], yd[MAXN];
    for (int i = 0; i < NP; ++i) { zd[i] = Dual<NP>(z[i]); zd[i].d[i] = 1.0; }
    evaluate(M, rule, sig, zd, t, dt, yd);
    for (int i = 0; i < n; ++i) { if (y) y[i] = yd[i].v; for (int j = 0; j < NP; ++j) J[i + j * n] = yd[i].d[j]; }
}
typedef void (*jac_ad_fn)(const OModel&, int, int, const double*, double, double, double*, double*);
template <int NP> struct JacTable { static void fill(jac_ad_fn* tab) { tab[NP] = &jac_forward_ad<NP>; JacTable<NP - 1>::fill(tab); } };
template <> struct JacTable<1> { static void fill(jac_ad_fn* tab) { tab[1] = &jac_forward_ad<1>; } };
static jac_ad_fn* jac_table() { static jac_ad_fn tab[MAXZ + 1]; static bool init = false; if (!init) { JacTable<MAXZ>::fill(tab); init = true; } return tab; }

// FiniteDiff forward differences, eps_i = max(sqrt(eps)*|z_i|, sqrt(eps)) (step rule unpinned)
static void jac_finite_diff(const OModel& M, int rule, int sig, const double* z, double t, double dt, double* J, double* y) {
    int n = M.n, nz = M.n + M.m; double f0[MAXN], f1[MAXN], z1[MAXZ];
    evaluate(M, rule, sig, z, t, dt, f0);
    const double se = std::sqrt(2.220446049250313e-16);
    for (int j = 0; j < nz; ++j) {
        for (int i = 0; i < nz; ++i) z1[i] = z[i];
        double eps = std::fmax(se * std::fabs(z[j]), se); z1[j] = z[j] + eps;
        evaluate(M, rule, sig, z1, t, dt, f1);
        for (int i = 0; i < n; ++i) J[i + j * n] = (f1[i] - f0[i]) / eps;
    }
    if (y) for (int i = 0; i < n; ++i) y[i] = f0[i];
}

// dense helpers for the chain rule (n x n, n x m; column-major)
static void mm(int r, int k, int c, const double* A, const double* B, double* C) {   // C = A(rxk) B(kxc)
    for (int j = 0; j < c; ++j) for (int i = 0; i < r; ++i) { double acc = 0; for (int l = 0; l < k; ++l) acc += A[i + l * r] * B[l + j * k]; C[i + j * r] = acc; }
}
// integrator-level UserDefined Jacobian: src/integration.jl:85-93 (Euler), :149-177 (RK3), :302-337 (RK4);
// RK2: A_d = I + A2 (I + A1 h/2) h, B_d = B2 h + A2 B1 h^2/2
static int jac_chain_rule(const OModel& M, int rule, const double* z, double t, double h, double* J, double* y) {
    int n = M.n, m = M.m, nz = n + m; const double* x = z; double u[MAXM];
    for (int i = 0; i < m; ++i) u[i] = (h == 0.0) ? 0.0 : z[n + i];
    std::vector<double> Jc(n * nz);
    if (rule < 0) { int rc = user_jacobian(M, x, u, t, J); if (y) dynamics(M, x, u, t, y); return rc; }
    int ns = (rule == RULE_EULER) ? 1 : (rule == RULE_RK2) ? 2 : (rule == RULE_RK3) ? 3 : 4;
    double k[4][MAXN], xs[MAXN];
    std::vector<double> A[4], B[4], dA[4], dB[4];
    for (int s = 0; s < ns; ++s) { A[s].resize(n * n); B[s].resize(n * m); dA[s].resize(n * n); dB[s].resize(n * m); }
    // stage points exactly as the StaticReturn methods build them
    for (int s = 0; s < ns; ++s) {
        double ts = t;
        if (s == 0) { for (int i = 0; i < n; ++i) xs[i] = x[i]; }
        else if (rule == RULE_RK2) { for (int i = 0; i < n; ++i) xs[i] = x[i] + k[0][i] / 2; ts = t + h / 2; }
        else if (rule == RULE_RK3) {
            if (s == 1) { for (int i = 0; i < n; ++i) xs[i] = x[i] + k[0][i] / 2; ts = t + h / 2; }
            else { for (int i = 0; i < n; ++i) xs[i] = x[i] - k[0][i] + 2 * k[1][i]; ts = t + h; }
        } else {
            if (s == 1) { for (int i = 0; i < n; ++i) xs[i] = x[i] + k[0][i] / 2; ts = t + h / 2; }
            else if (s == 2) { for (int i = 0; i < n; ++i) xs[i] = x[i] + k[1][i] / 2; ts = t + h / 2; }
            else { for (int i = 0; i < n; ++i) xs[i] = x[i] + k[2][i]; ts = t + h; }
        }
        double f[MAXN]; dynamics(M, xs, u, ts, f); for (int i = 0; i < n; ++i) k[s][i] = f[i] * h;
        int rc = user_jacobian(M, xs, u, ts, Jc.data()); if (rc) return rc;
        for (int i = 0; i < n * n; ++i) A[s][i] = Jc[i]; for (int i = 0; i < n * m; ++i) B[s][i] = Jc[n * n + i];
    }
    std::vector<double> T1(n * n), T2(n * m), W(n * n), Wb(n * m);
    auto eye_plus = [&](double c1, const std::vector<double>* d1, double c2, const std::vector<double>* d2, std::vector<double>& out) {
        for (int i = 0; i < n * n; ++i) out[i] = (d1 ? c1 * (*d1)[i] : 0.0) + (d2 ? c2 * (*d2)[i] : 0.0);
        for (int i = 0; i < n; ++i) out[i + i * n] += 1.0; };
    for (int i = 0; i < n * n; ++i) dA[0][i] = A[0][i] * h;
    for (int i = 0; i < n * m; ++i) dB[0][i] = B[0][i] * h;
    if (rule == RULE_EULER) {
        for (int i = 0; i < n * n; ++i) J[i] = dA[0][i]; for (int i = 0; i < n; ++i) J[i + i * n] += 1.0;
        for (int i = 0; i < n * m; ++i) J[n * n + i] = dB[0][i];
    } else if (rule == RULE_RK2) {
        eye_plus(0.5, &dA[0], 0, nullptr, W); mm(n, n, n, A[1].data(), W.data(), T1.data());
        mm(n, n, m, A[1].data(), dB[0].data(), T2.data());
        for (int i = 0; i < n * n; ++i) J[i] = T1[i] * h; for (int i = 0; i < n; ++i) J[i + i * n] += 1.0;
        for (int i = 0; i < n * m; ++i) J[n * n + i] = B[1][i] * h + 0.5 * T2[i] * h;
    } else if (rule == RULE_RK3) {
        eye_plus(0.5, &dA[0], 0, nullptr, W); mm(n, n, n, A[1].data(), W.data(), T1.data()); for (int i = 0; i < n * n; ++i) dA[1][i] = T1[i] * h;
        eye_plus(-1.0, &dA[0], 2.0, &dA[1], W); mm(n, n, n, A[2].data(), W.data(), T1.data()); for (int i = 0; i < n * n; ++i) dA[2][i] = T1[i] * h;
        mm(n, n, m, A[1].data(), dB[0].data(), T2.data()); for (int i = 0; i < n * m; ++i) dB[1][i] = B[1][i] * h + 0.5 * T2[i] * h;
        for (int i = 0; i < n * m; ++i) Wb[i] = 2 * dB[1][i] - dB[0][i];
        mm(n, n, m, A[2].data(), Wb.data(), T2.data()); for (int i = 0; i < n * m; ++i) dB[2][i] = B[2][i] * h + T2[i] * h;
        for (int i = 0; i < n * n; ++i) J[i] = (dA[0][i] + 4 * dA[1][i] + dA[2][i]) / 6; for (int i = 0; i < n; ++i) J[i + i * n] += 1.0;
        for (int i = 0; i < n * m; ++i) J[n * n + i] = (dB[0][i] + 4 * dB[1][i] + dB[2][i]) / 6;
    } else {
        eye_plus(0.5, &dA[0], 0, nullptr, W); mm(n, n, n, A[1].data(), W.data(), T1.data()); for (int i = 0; i < n * n; ++i) dA[1][i] = T1[i] * h;
        eye_plus(0.5, &dA[1], 0, nullptr, W); mm(n, n, n, A[2].data(), W.data(), T1.data()); for (int i = 0; i < n * n; ++i) dA[2][i] = T1[i] * h;
        eye_plus(1.0, &dA[2], 0, nullptr, W); mm(n, n, n, A[3].data(), W.data(), T1.data()); for (int i = 0; i < n * n; ++i) dA[3][i] = T1[i] * h;
        mm(n, n, m, A[1].data(), dB[0].data(), T2.data()); for (int i = 0; i < n * m; ++i) dB[1][i] = B[1][i] * h + 0.5 * T2[i] * h;
        mm(n, n, m, A[2].data(), dB[1].data(), T2.data()); for (int i = 0; i < n * m; ++i) dB[2][i] = B[2][i] * h + 0.5 * T2[i] * h;
        mm(n, n, m, A[3].data(), dB[2].data(), T2.data()); for (int i = 0; i < n * m; ++i) dB[3][i] = B[3][i] * h + T2[i] * h;
        for (int i = 0; i < n * n; ++i) J[i] = (dA[0][i] + 2 * dA[1][i] + 2 * dA[2][i] + dA[3][i]) / 6; for (int i = 0; i < n; ++i) J[i + i * n] += 1.0;
        for (int i = 0; i < n * m; ++i) J[n * n + i] = (dB[0][i] + 2 * dB[1][i] + 2 * dB[2][i] + dB[3][i]) / 6;
    }
    if (h == 0.0) for (int i = 0; i < n * m; ++i) J[n * n + i] = 0.0;   // terminal knot: control is `false * u`
    if (y) { integrate(M, rule, SIG_STATIC_RETURN, x, u, t, h, y); }
    return 0;
}

// ------------------------------------------------------------------------------------------
// LieState index algebra (src/liestate.jl:121-131) and error-state functions
// ------------------------------------------------------------------------------------------
static int rot_params(int rot) { return rot == ROT_QUAT ? 4 : 3; }   // src/liestate.jl:42-46
struct OLie { int rot; int nP; int P[8]; };
static int lie_len(const OLie& s) { int a = 0; for (int i = 0; i < s.nP; ++i) a += s.P[i]; return rot_params(s.rot) * (s.nP - 1) + a; }
static int lie_errdim(const OLie& s) { int a = 0; for (int i = 0; i < s.nP; ++i) a += s.P[i]; return 3 * (s.nP - 1) + a; }
// 1-based first index of rotation i (1-based)  ==  first(rot_inds(R,P,i))
static int lie_rot_first(const OLie& s, int i) { int a = 0; for (int k = 0; k < i; ++k) a += s.P[k]; return a + (i - 1) * rot_params(s.rot) + 1; }
static int lie_vec_first(const OLie& s, int i) { int a = 0; for (int k = 0; k < i - 1; ++k) a += s.P[k]; return a + (i - 1) * rot_params(s.rot) + 1; }

static int model_lie(const OModel& M, OLie* s) {
    s->rot = M.rot;
    switch (M.kind) {
        case M_SATELLITE: case M_QUADROTOR: case M_TV_SATELLITE: s->nP = 2; s->P[0] = 3; s->P[1] = 6; return 1;   // src/rigidbody.jl:48
        case M_SATELLITE_PAIR: s->nP = 3; s->P[0] = 0; s->P[1] = 3; s->P[2] = 3; return 1;     // examples/satellite_pair.jl:40
        case M_LIE_SATELLITE: s->nP = 2; s->P[0] = 3; s->P[1] = 0; return 1;                  // test/liemodel.jl:12
        case M_LIE_STATE: s->nP = (int)M.p[0]; for (int i = 0; i < s->nP && i < 8; ++i) s->P[i] = (int)M.p[1 + i]; return s->nP >= 1 && s->nP <= 8;
        default: return 0;   // EuclideanState
    }
}
// QuatRotation(w,x,y,z) constructor normalises by default (Rotations.jl 1.x); used by
// state_diff / errstate_jacobian! / nabla-errstate (src/liestate.jl:216-217, :269, :307)
static void qnormalize(const double* q, double out[4]) {
    double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (int i = 0; i < 4; ++i) out[i] = q[i] / nn;
}

extern "C" {

int oracle_model_init(OModel* M, int kind, int rot, int frame, const double* params, int nparams) {
    std::memset(M, 0, sizeof(OModel)); M->kind = kind; M->rot = rot; M->frame = frame; M->np = nparams;
    if (nparams > 64) return -1; for (int i = 0; i < nparams; ++i) M->p[i] = params[i];
    int rp = rot_params(rot);
    switch (kind) {
        case M_CARTPOLE: M->n = 4; M->m = 1; break;
        case M_DOUBLE_INTEGRATOR: M->m = (int)params[0]; M->n = 2 * M->m; break;
        case M_LINEAR: M->n = (int)params[0]; M->m = (int)params[1]; break;
        case M_SATELLITE: M->n = 9 + rp; M->m = 6; break;
        case M_QUADROTOR: M->n = 9 + rp; M->m = 4; break;
        case M_SATELLITE_PAIR: M->n = 2 * rp + 6; M->m = 6; break;
        case M_LIE_SATELLITE: M->n = 3 + rp; M->m = 3; break;
        case M_TV_PENDULUM: M->n = 2; M->m = 1; break;
        case M_TV_SATELLITE: M->n = 9 + rp; M->m = 6; break;
        case M_LIE_STATE: { OLie ls; if (!model_lie(*M, &ls)) return -1; M->n = lie_len(ls); M->m = 0; break; }
        default: return -1;
    }
    if (M->n > MAXN || M->m > MAXM) return -1;
    return 0;
}
int oracle_sizeof_model() { return (int)sizeof(OModel); }

// dims(model) = (n, m, p) and errstate_dim (src/functionbase.jl:130-135, src/liestate.jl:24,124)
int oracle_dims(const OModel* M, int* n, int* m, int* p, int* e) {
    *n = M->n; *m = M->m; *p = M->n; OLie s; *e = model_lie(*M, &s) ? lie_errdim(s) : M->n; return 0;
}
// LieState queries for an arbitrary (rot, P): length, errstate_dim, rot_inds/vec_inds first indices (1-based)
int oracle_liestate_query(int rot, const int* P, int nP, int* len, int* edim, int* rot_first, int* vec_first) {
    OLie s; s.rot = rot; s.nP = nP; for (int i = 0; i < nP; ++i) s.P[i] = P[i];
    *len = lie_len(s); *edim = lie_errdim(s);
    for (int i = 1; i <= nP - 1; ++i) rot_first[i - 1] = lie_rot_first(s, i);
    for (int i = 1; i <= nP; ++i) vec_first[i - 1] = lie_vec_first(s, i);
    return 0;
}
int oracle_model_liestate(const OModel* M, int* rot, int* nP, int* P) {
    OLie s; if (!model_lie(*M, &s)) { *nP = 0; return 0; } *rot = s.rot; *nP = s.nP; for (int i = 0; i < s.nP; ++i) P[i] = s.P[i]; return 0;
}
// gen_inds (src/rigidbody.jl:80-94): 1-based first index and length of r,q,v,w,u
int oracle_gen_inds(const OModel* M, int* first, int* len) {
    if (M->kind != M_SATELLITE && M->kind != M_QUADROTOR) return -1;
    int rp = rot_params(M->rot);
    first[0] = 1; len[0] = 3; first[1] = 4; len[1] = rp; first[2] = 4 + rp; len[2] = 3; first[3] = 7 + rp; len[3] = 3; first[4] = 10 + rp; len[4] = M->m; return 0;
}

// Z: (n+m) x N column-major; t, dt: N
int oracle_dynamics(const OModel* M, int N, const double* Z, const double* t, const double* dt, double* Xdot) {
    if (M->rot != ROT_QUAT && M->kind >= M_SATELLITE && M->kind != M_SATELLITE && M->kind != M_QUADROTOR) return -2;   // three-parameter attitudes: the RigidBody satellite / quadrotor only
    int nz = M->n + M->m;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < N; ++k) evaluate(*M, -1, 0, Z + (size_t)k * nz, t[k], dt[k], Xdot + (size_t)k * M->n);
    return 0;
}
int oracle_discrete_dynamics(const OModel* M, int rule, int sig, int N, const double* Z, const double* t, const double* dt, double* Xn) {
    if (M->rot != ROT_QUAT && M->kind >= M_SATELLITE && M->kind != M_SATELLITE && M->kind != M_QUADROTOR) return -2;   // three-parameter attitudes: the RigidBody satellite / quadrotor only
    int nz = M->n + M->m;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < N; ++k) evaluate(*M, rule, sig, Z + (size_t)k * nz, t[k], dt[k], Xn + (size_t)k * M->n);
    return 0;
}
// rule < 0: continuous jacobian!(sig,diff,model,J,y,z); else DiscretizedDynamics{rule}
int oracle_jacobian(const OModel* M, int rule, int diff, int sig, int N, const double* Z, const double* t, const double* dt, double* J, double* Y) {
    if (M->rot != ROT_QUAT && M->kind >= M_SATELLITE && M->kind != M_SATELLITE && M->kind != M_QUADROTOR) return -2;   // three-parameter attitudes: the RigidBody satellite / quadrotor only
    int n = M->n, nz = M->n + M->m; int rc_all = 0; jac_ad_fn* tab = jac_table();
#pragma omp parallel for schedule(static)
    for (int k = 0; k < N; ++k) {
        const double* z = Z + (size_t)k * nz; double* Jk = J + (size_t)k * n * nz; double* yk = Y ? Y + (size_t)k * n : nullptr;
        if (diff == DIFF_FORWARD_AD) tab[nz](*M, rule, sig, z, t[k], dt[k], Jk, yk);
        else if (diff == DIFF_FINITE_DIFF) jac_finite_diff(*M, rule, sig, z, t[k], dt[k], Jk, yk);
        else { int rc = jac_chain_rule(*M, rule, z, t[k], dt[k], Jk, yk); if (rc) rc_all = rc; }
    }
    return rc_all;
}

// wrenches(model, x, u, t) per knot point (W: 6 x N) and its Jacobian with respect to z (F: 6 x (n+m) x N), RigidBody models only
int oracle_wrenches(const OModel* M, int N, const double* Z, const double* t, double* W) {
    int nz = M->n + M->m; int rc_all = 0;
    for (int k = 0; k < N; ++k) { const double* z = Z + (size_t)k * nz; int rc = wrench_any(*M, z, z + M->n, t[k], W + (size_t)k * 6); if (rc) rc_all = rc; }
    return rc_all;
}
int oracle_wrench_jacobian(const OModel* M, int N, const double* Z, const double* t, double* F) {
    static wrench_jac_fn tab[MAXZ + 1]; static bool init = false;
    if (!init) { WrenchTable<MAXZ>::fill(tab); init = true; }
    int nz = M->n + M->m; int rc_all = 0;
    for (int k = 0; k < N; ++k) { int rc = tab[nz](*M, Z + (size_t)k * nz, t[k], F + (size_t)k * 6 * nz, nullptr); if (rc) rc_all = rc; }
    return rc_all;
}

// errstate_jacobian!(model, G, x): src/statevectortype.jl:149-155 (Euclidean: G = I, cleared),
// src/liestate.jl:262-298 (Lie: ones on vector diagonals + L(q)H blocks; G is NOT cleared unless clear!=0)
// X: ldx x N (state in rows 0..n-1), G: n x e x N column-major
int oracle_errstate_jacobian(const OModel* M, int N, const double* X, int ldx, int clear, double* G) {
    OLie s; int lie = model_lie(*M, &s); int n = M->n; int e = lie ? lie_errdim(s) : n;
    for (int k = 0; k < N; ++k) {
        const double* x = X + (size_t)k * ldx; double* Gk = G + (size_t)k * n * e;
        if (!lie) { for (int i = 0; i < n * e; ++i) Gk[i] = 0; for (int i = 0; i < n; ++i) Gk[i + i * n] = 1.0; continue; }
        if (clear) for (int i = 0; i < n * e; ++i) Gk[i] = 0;
        int r = 0, c = 0;
        for (int v = 0; v < s.nP; ++v) {
            for (int j = 0; j < s.P[v]; ++j) { Gk[r + c * n] = 1.0; ++r; ++c; }
            if (v < s.nP - 1 && s.rot != ROT_QUAT) {                              // grad-differential(p), 3 x 3
                double A[3][3]; rot3_gradient(s.rot, x + r, A);
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Gk[(r + i) + (c + j) * n] = A[i][j];
                r += 3; c += 3;
            } else if (v < s.nP - 1) {
                double q[4]; qnormalize(x + r, q); double L[4][4]; lmult(q, L);   // grad-differential(q) = L(q) H
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) Gk[(r + i) + (c + j) * n] = L[i][j + 1];
                r += 4; c += 3;
            }
        }
    }
    return 0;
}
// inv(errmap)(dq) of Rotations.jl 1.x [third party, restated from its published definitions; parity unpinned -- only CayleyMap's value is pinned by the
// reference's tests, test/rbstate.jl:157], dq = [w; v] the quaternion of R0 \ R.  scaling(ExponentialMap) = 1/2, scaling(MRPMap) = 2, others 1:
//   InvExponentialMap  0.5 logm(dq),  logm(dq) = 2 vector(_log_as_quat(dq)),  _log_as_quat: M = atan(theta, w) / theta for theta = |v| > 1e-6,
//                      else (1 - theta^2 / (3 w^2)) / w;  the result is M v
//   InvCayleyMap       v / w            InvMRPMap  2 v / (1 + w)            InvQuatVecMap  v sign(w)
//   InvIdentityMap     params(R0 \ R): three values for MRP (v / (1 + w)) and RodriguesParam (v / w) states; four for a quaternion, which
//                      Rotations.rotation_error cannot wrap in its three-element RotationError (it throws): rc -3 here
static int errmap_inverse(int errmap, int rot, const double dq[4], double out[3]) {
    const double w = dq[0]; const double* v = dq + 1; double M;
    switch (errmap) {
        case 0: M = 1.0 / w; for (int j = 0; j < 3; ++j) out[j] = v[j] / w; return 0;
        case 1: { const double th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); M = th > 1e-6 ? std::atan2(th, w) / th : (1.0 - th * th / (3.0 * w * w)) / w; break; }
        case 2: M = 2.0 / (1.0 + w); break;
        case 3: M = w > 0 ? 1.0 : (w < 0 ? -1.0 : 0.0); break;
        case 4: if (rot == ROT_QUAT) return -3; M = rot == ROT_MRP ? 1.0 / (1.0 + w) : 1.0 / w; break;
        default: return -1;
    }
    for (int j = 0; j < 3; ++j) out[j] = v[j] * M;
    return 0;
}
// state_diff(model, x, x0[, errmap]) (src/liestate.jl:26-32,210-260): Euclidean x - x0; rotations: rotation_error(R, R0, errmap) = inv(errmap)(R0 \ R)
// (:216-217); errmap 0 CayleyMap (the default) 1 ExponentialMap 2 MRPMap 3 QuatVecMap 4 IdentityMap
int oracle_state_diff_map(const OModel* M, int N, const double* X, int ldx, const double* X0, int ldx0, int errmap, double* dX) {
    OLie s; int lie = model_lie(*M, &s); int n = M->n; int e = lie ? lie_errdim(s) : n;
    for (int k = 0; k < N; ++k) {
        const double* x = X + (size_t)k * ldx; const double* x0 = X0 + (size_t)k * ldx0; double* d = dX + (size_t)k * e;
        if (!lie) { for (int i = 0; i < n; ++i) d[i] = x[i] - x0[i]; continue; }
        int r = 0, c = 0;
        for (int v = 0; v < s.nP; ++v) {
            for (int j = 0; j < s.P[v]; ++j) { d[c] = x[r] - x0[r]; ++r; ++c; }
            if (v < s.nP - 1) {
                double q[4], q0[4];
                if (s.rot == ROT_QUAT) { qnormalize(x + r, q); qnormalize(x0 + r, q0); }
                else { rot3_quat(s.rot, x + r, q); rot3_quat(s.rot, x0 + r, q0); }      // inv(errmap)(R0 \ R) goes through the quaternion of R0 \ R
                double q0c[4] = {q0[0], -q0[1], -q0[2], -q0[3]}; double dq[4]; qmul(q0c, q, dq);
                const int rc = errmap_inverse(errmap, s.rot, dq, d + c); if (rc) return rc;
                r += rot_params(s.rot); c += 3;
            }
        }
    }
    return 0;
}
int oracle_state_diff(const OModel* M, int N, const double* X, int ldx, const double* X0, int ldx0, double* dX) {
    return oracle_state_diff_map(M, N, X, ldx, X0, ldx0, 0, dX);
}
// grad^2-differential(p, b) of the three-parameter rotations [3P, Rotations.jl; restated from its definition]: the contraction with b of the second
// derivative of the composition map delta -> params(p * R(delta)) at delta = 0 (delta in the rotation's own parameters) -- the derivative with respect
// to delta of grad-differential(p * R(delta))' b.  From the composition laws, with a = p.p:
//   MRP             (1 - |d|^2) p + (1 - a) d + 2 p x d  over  1 + a |d|^2 - 2 p.d
//                   second order terms: p (4 (p.d)^2 - (1 + a) |d|^2) + 2 (p.d) A0 d,   A0 = (1 - a) I + 2 skew(p)
//                   =>  (b.p) (8 p p' - 2 (1 + a) I) + 2 (p c' + c p'),   c = A0' b = (1 - a) b - 2 p x b
//   RodriguesParam  g + d + g x d  over  1 - g.d
//                   second order terms: g (g.d)^2 + (g.d) B0 d,   B0 = I + skew(g)
//                   =>  2 (b.g) g g' + g c' + c g',   c = B0' b = b - g x b
// (for a quaternion the same definition with the Cayley map gives -(q.b) I: the unit test of test/liestate.jl:93-99 and the formula below).
// Held against tests/golden/make_golden_rot3.py, which differentiates the composition twice by automatic differentiation.
static void rot3_hessian(int rot, const double p[3], const double b[3], double H[3][3]) {
    const double a = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], bp = b[0] * p[0] + b[1] * p[1] + b[2] * p[2];
    const double pxb[3] = {p[1] * b[2] - p[2] * b[1], p[2] * b[0] - p[0] * b[2], p[0] * b[1] - p[1] * b[0]};
    double c[3];
    for (int i = 0; i < 3; ++i) c[i] = rot == ROT_MRP ? (1.0 - a) * b[i] - 2.0 * pxb[i] : b[i] - pxb[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        if (rot == ROT_MRP) H[i][j] = bp * (8.0 * p[i] * p[j] - ((i == j) ? 2.0 * (1.0 + a) : 0.0)) + 2.0 * (p[i] * c[j] + c[i] * p[j]);
        else H[i][j] = 2.0 * bp * p[i] * p[j] + p[i] * c[j] + c[i] * p[j];
    }
}
// nabla-errstate_jacobian!(model, nablaG, x, xbar): Euclidean -> 0 ; Lie: 3x3 diagonal blocks grad^2-differential(R(x_rot), xbar_rot): -(q . xbar_q) I for
// quaternions (src/liestate.jl:300-320; only those blocks are written unless clear != 0)
int oracle_nabla_errstate_jacobian(const OModel* M, int N, const double* X, int ldx, const double* Xb, int ldxb, int clear, double* dG) {
    OLie s; int lie = model_lie(*M, &s); int n = M->n; int e = lie ? lie_errdim(s) : n;
    if (lie && s.rot != ROT_QUAT) {
        for (int k = 0; k < N; ++k) {
            const double* x = X + (size_t)k * ldx; const double* xb = Xb + (size_t)k * ldxb; double* Hk = dG + (size_t)k * e * e;
            if (clear) for (int i = 0; i < e * e; ++i) Hk[i] = 0;
            int r = 0, c = 0;
            for (int v = 0; v < s.nP; ++v) {
                r += s.P[v]; c += s.P[v];
                if (v < s.nP - 1) {
                    double H[3][3]; rot3_hessian(s.rot, x + r, xb + r, H);
                    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Hk[(c + i) + (c + j) * e] = H[i][j];
                    r += 3; c += 3;
                }
            }
        }
        return 0;
    }
    for (int k = 0; k < N; ++k) {
        const double* x = X + (size_t)k * ldx; const double* xb = Xb + (size_t)k * ldxb; double* Hk = dG + (size_t)k * e * e;
        if (!lie) { for (int i = 0; i < e * e; ++i) Hk[i] = 0; continue; }
        if (clear) for (int i = 0; i < e * e; ++i) Hk[i] = 0;
        int r = 0, c = 0;
        for (int v = 0; v < s.nP; ++v) {
            r += s.P[v]; c += s.P[v];
            if (v < s.nP - 1) {
                double q[4]; qnormalize(x + r, q); double dot = 0; for (int i = 0; i < 4; ++i) dot += q[i] * xb[r + i];
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Hk[(c + i) + (c + j) * e] = (i == j) ? -dot : 0.0;
                r += 4; c += 3;
            }
        }
    }
    return 0;
}
// rollout!(sig, model, Z, x0): src/trajectories.jl:436-441. Z: (n+m) x K x Mtraj, states overwritten in place
int oracle_rollout(const OModel* M, int rule, int sig, int K, int ntraj, double* Z, const double* t, const double* dt, const double* X0) {
    int n = M->n, nz = M->n + M->m;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ntraj; ++j) {
        double* Zj = Z + (size_t)j * K * nz; const double* tj = t + (size_t)j * K; const double* dtj = dt + (size_t)j * K;
        if (X0) for (int i = 0; i < n; ++i) Zj[i] = X0[(size_t)j * n + i];
        for (int k = 1; k < K; ++k) evaluate(*M, rule, sig, Zj + (size_t)(k - 1) * nz, tj[k - 1], dtj[k - 1], Zj + (size_t)k * nz);
    }
    return 0;
}
// ---- ImplicitMidpoint (src/integration.jl:620-729) with the Newton solve of src/integration.jl:422-463 --------------------
//   e(x2; z1) = x1 + h f((x1 + x2)/2, u1, t + h/2) - x2 = 0      J2 = de/dz2 = [(h/2) A - I, 0]      J1 = de/dz1 = [I + (h/2) A, h B]
// with [A B] the continuous Jacobian at (xmid, u1) (src/integration.jl:664-690).  Newton from x2 = x1 (StaticReturn flavour):
// residual, Jacobians, convergence test, step -- so on exit J2/J1 belong to the last iterate that was TESTED; the discrete Jacobian
// is the implicit-function-theorem solve  J = -J2[:, 1:n] \ J1  (src/integration.jl:517-538).  LU with partial pivoting as `lu`.
static int lu_factor(double* A, int n, int* piv) {        // A column-major n x n, in place
    for (int k = 0; k < n; ++k) {
        int p = k; double best = std::fabs(A[k + k * n]);
        for (int i = k + 1; i < n; ++i) if (std::fabs(A[i + k * n]) > best) { best = std::fabs(A[i + k * n]); p = i; }
        piv[k] = p;
        if (p != k) for (int j = 0; j < n; ++j) std::swap(A[k + j * n], A[p + j * n]);
        if (A[k + k * n] == 0.0) return -1;
        for (int i = k + 1; i < n; ++i) {
            A[i + k * n] /= A[k + k * n];
            for (int j = k + 1; j < n; ++j) A[i + j * n] -= A[i + k * n] * A[k + j * n];
        }
    }
    return 0;
}
static void lu_solve(const double* LU, const int* piv, int n, double* b) {
    for (int k = 0; k < n; ++k) if (piv[k] != k) std::swap(b[k], b[piv[k]]);
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) b[i] -= LU[i + j * n] * b[j];
    for (int i = n - 1; i >= 0; --i) { for (int j = i + 1; j < n; ++j) b[i] -= LU[i + j * n] * b[j]; b[i] /= LU[i + i * n]; }
}
// residual and Jacobians of the implicit midpoint dynamics error for one pair; Jc: n x nz continuous Jacobian scratch
static void implicit_midpoint_parts(const OModel& M, const double* x1, const double* u1, const double* x2, double t, double h,
                                    double* e, double* J2, double* J1) {
    const int n = M.n, nz = M.n + M.m;
    double zm[MAXZ], Jc[MAXN * MAXZ], f[MAXN];
    for (int i = 0; i < n; ++i) zm[i] = (x1[i] + x2[i]) / 2;
    for (int i = 0; i < M.m; ++i) zm[n + i] = u1[i];
    // the residual evaluates the dynamics at t + h/2 (src/integration.jl:640,659), its Jacobians take the continuous Jacobian at t (:684,:716)
    jac_table()[nz](M, -1, 0, zm, t, h, Jc, f);
    dynamics(M, zm, zm + n, t + h / 2, f);
    for (int i = 0; i < n; ++i) e[i] = x1[i] + h * f[i] - x2[i];
    for (int j = 0; j < nz; ++j) for (int i = 0; i < n; ++i) {
        const double a = Jc[i + j * n] * h;
        if (j < n) { J1[i + j * n] = a / 2 + (i == j ? 1.0 : 0.0); J2[i + j * n] = a / 2 - (i == j ? 1.0 : 0.0); }
        else { J1[i + j * n] = a; J2[i + j * n] = 0.0; }
    }
}
// Xn: n x N ; J: n x nz x N or NULL ; iters_used: N or NULL
int oracle_implicit_midpoint(const OModel* M, int N, const double* Z, const double* t, const double* dt, int newton_iters, double tol,
                             double* Xn, double* J, int* iters_used) {
    if (M->rot != ROT_QUAT && M->kind >= M_SATELLITE && M->kind != M_SATELLITE && M->kind != M_QUADROTOR) return -2;   // three-parameter attitudes: the RigidBody satellite / quadrotor only
    const int n = M->n, nz = M->n + M->m;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < N; ++k) {
        const double* z = Z + (size_t)k * nz; const double h = dt[k];
        double u[MAXM]; for (int i = 0; i < M->m; ++i) u[i] = (h == 0.0) ? 0.0 : z[n + i];       // terminal knot point: control reads as zeros
        double xn[MAXN], e[MAXN], J2[MAXN * MAXZ], J1[MAXN * MAXZ], A[MAXN * MAXN]; int piv[MAXN];
        for (int i = 0; i < n; ++i) xn[i] = z[i];
        int it = 0;
        for (; it < newton_iters; ++it) {
            implicit_midpoint_parts(*M, z, u, xn, t[k], h, e, J2, J1);
            double nr = 0; for (int i = 0; i < n; ++i) nr += e[i] * e[i];
            if (std::sqrt(nr) < tol) break;
            for (int i = 0; i < n * n; ++i) A[i] = J2[i];
            lu_factor(A, n, piv); lu_solve(A, piv, n, e);
            for (int i = 0; i < n; ++i) xn[i] -= e[i];
        }
        for (int i = 0; i < n; ++i) Xn[(size_t)k * n + i] = xn[i];
        if (iters_used) iters_used[k] = it;
        if (J) {
            for (int i = 0; i < n * n; ++i) A[i] = J2[i];
            lu_factor(A, n, piv);
            for (int j = 0; j < nz; ++j) {
                double col[MAXN]; for (int i = 0; i < n; ++i) col[i] = J1[i + j * n];
                lu_solve(A, piv, n, col);
                const bool zero = (j >= n && h == 0.0);
                for (int i = 0; i < n; ++i) J[(size_t)k * n * nz + i + j * n] = zero ? 0.0 : -col[i];
            }
        }
    }
    return 0;
}
// dynamics_error / dynamics_error_jacobian! of ImplicitMidpoint for the N-1 consecutive pairs: E n x (N-1), J2, J1: n x nz x (N-1) (any may be NULL)
int oracle_implicit_midpoint_error(const OModel* M, int N, const double* Z, const double* t, const double* dt, double* E, double* J2, double* J1) {
    if (M->rot != ROT_QUAT && M->kind >= M_SATELLITE && M->kind != M_SATELLITE && M->kind != M_QUADROTOR) return -2;   // three-parameter attitudes: the RigidBody satellite / quadrotor only
    const int n = M->n, nz = M->n + M->m;
    for (int k = 0; k + 1 < N; ++k) {
        const double* z1 = Z + (size_t)k * nz; const double* z2 = Z + (size_t)(k + 1) * nz;
        double u[MAXM]; for (int i = 0; i < M->m; ++i) u[i] = (dt[k] == 0.0) ? 0.0 : z1[n + i];
        double e[MAXN], j2[MAXN * MAXZ], j1[MAXN * MAXZ];
        implicit_midpoint_parts(*M, z1, u, z2, t[k], dt[k], e, j2, j1);
        if (E) for (int i = 0; i < n; ++i) E[(size_t)k * n + i] = e[i];
        if (J2) for (int i = 0; i < n * nz; ++i) J2[(size_t)k * n * nz + i] = j2[i];
        if (J1) for (int i = 0; i < n * nz; ++i) J1[(size_t)k * n * nz + i] = j1[i];
    }
    return 0;
}

int oracle_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_num_threads(int nt) {
#ifdef _OPENMP
    omp_set_num_threads(nt);
#else
    (void)nt;
#endif
}

}  // extern "C"
