// fam_rigid.hpp -- shared by the rigid-body translation units: model aliases, column plans and the per-variant entry points.
// Each (wrench, velocity frame, inertia structure) variant is its own translation unit so that the compile-time-column
// ForwardAD kernels (one instantiation per rule and dtype, ~10-20 s each) build in parallel.
#pragma once
#include "dispatch.hpp"
namespace rd {
// Satellite columns: r 0-2, q 3-6, v 7-9, w 10-12, uF 13-15, tau 16-18.  Measured fp64 instruction counts per column of the RK4
// tangent (diagonal J): r 0, v 2, q ~225, w ~343, uF ~57, tau ~257 (+ ~150-200 shared per wave).  Two constraints shape the table: the four
// waves carry similar totals (~740-830), and every wave carries ONE quaternion column ahead of everything expensive, so that the leading ten
// columns (r, q, v: 53 % of a row) of a workgroup's first tile are final after a quarter of the arithmetic and their whole 128-byte lines
// leave while the rest is computed (TablePlanS, kernels.hpp; tuning harness, back-to-back launches at N = 65 536: 28.1 -> 26.4 us).
//   wave 0: r, q3, w10, uF13   wave 1: q4, v, w11, uF14, uF15   wave 2: q5, w12, tau18   wave 3: q6, tau16, tau17 (+ f(z))
// (round 3: f(z) from wave 3 instead of wave 1 -- the value wave keeps the primal outputs live through its first column, and on wave 1 that cost 12 bytes
// of scratch per lane: 26.85 -> 26.15 us single stream in the same-box A/B of tune/tune_plan.hip, body-frame velocities 51.5 -> 47.7, other rules equal)
using SatPlan = TablePlanS<19, 4, 3, 10, /*r*/ 0, 0, 0, /*q*/ 0, 1, 2, 3, /*v*/ 1, 1, 1, /*w*/ 0, 1, 2, /*uF*/ 0, 1, 1, /*tau*/ 3, 3, 2>;
// Quadrotor columns: r 0-2, q 3-6, v 7-9, w 10-12, u 13-16 (each rotor thrust feeds both F and tau, ~310 per column)
//   wave 0: q3, w10, u15   wave 1: q4, w11, u16   wave 2: r, q5, v, w12 (+ f(z))   wave 3: q6, u13, u14
using QuadPlan = TablePlanS<17, 4, 2, 10, /*r*/ 2, 2, 2, /*q*/ 0, 1, 2, 3, /*v*/ 2, 2, 2, /*w*/ 0, 1, 2, /*u*/ 3, 3, 0, 1>;
// every rule of the rigid bodies gets the compile-time-column (structural zeros folded) split-role kernel
template <class T, bool B, bool D, bool GZ, int RULE> struct UnrollPolicy<RigidBody<T, WRENCH_SATELLITE, B, D, GZ>, RULE> { static constexpr bool value = true; using Plan = SatPlan; static constexpr int NC = 4; };
template <class T, bool D, bool GZ, int RULE> struct UnrollPolicy<RigidBody<T, WRENCH_QUADROTOR, false, D, GZ>, RULE> { static constexpr bool value = true; using Plan = QuadPlan; static constexpr int NC = 4; };
// body-frame velocities: the same column assignment with f(z) from wave 3 (round 4, tune/tune_plan.hip -DQUAD -DBODYF=true: 32.7 -> 31.9 us; every plan tried
// spills 170-370 bytes per lane there -- a body-frame rotor column is 685 fp64 instructions on its own)
// (round 6: in the body frame the v columns are no longer free -- r' = R(q) v, v' = .. - w x v -- so they are dealt out, and one rotor column moves to the
// wave that had them; with the copy-wave prefetch and paced stores below: 32.0 -> 28.9 us at 65 536 knots, same box, tune/tune_plan.hip -DQUAD -DBODYF=true)
using QuadBodyPlan = TablePlanS<17, 4, 3, 10, /*r*/ 2, 2, 2, /*q*/ 0, 1, 2, 3, /*v*/ 3, 0, 1, /*w*/ 0, 1, 2, /*u*/ 3, 2, 0, 1>;
// The fp64 body-frame quadrotor's compute role does not fit 256 registers.  What the spills cost was not their count but WHICH values went: the next tile's
// prefetched inputs -- to spill a loaded value the wave first waits for it, i.e. for a memory round trip under the launch's saturated store stream, at the
// top of every tile (s_waitcnt vmcnt(0) in front of six scratch stores: the waves were parked 57 % of their cycles).  Compute-only 23.3 us, stores-only
// 22.8 us, together 32 us.  So the COPY waves prefetch for this model (kernels.hpp CopyPrefetchOf: scratch 172 -> 64 bytes per lane) and pace their stores
// (3 sleep units per store: the compute waves' remaining scratch accesses share the CU's memory path with them).  Measured on the satellites too, where
// both changes LOSE (world frame 26.9 -> 28.9 us, body frame 29.2 -> 30.2: their compute roles hold their prefetch, and the copy role spills instead).
template <bool D, bool GZ> struct CopyPrefetchModel<RigidBody<double, WRENCH_QUADROTOR, true, D, GZ>, double> { static constexpr bool value = true; static constexpr int PACE = 3; };
// FiniteDifference, fp64 body-frame satellite (92 B of scratch, the same stall): 32.2 -> 29.6 us with the copy-wave prefetch (0.57 -> 0.62); the body-frame
// quadrotor's FiniteDifference kernel spills 12 B and gains nothing, the world-frame kernels lose (28.4 -> 29.4 us): tune/tune_plan.hip -DFDK
template <bool D, bool GZ> struct CopyPrefetchFd<RigidBody<double, WRENCH_SATELLITE, true, D, GZ>, double> { static constexpr bool value = true; };
// UserDefined (chain rule on the analytic stage Jacobians), fp64 body frame: 180 B of scratch in both.  Satellite 38.1 -> 36.4 us with the prefetch (pace loses),
// quadrotor 39.1 -> 35.6 with the prefetch, 33.8 with pace 3 on top (tune/tune_plan.hip -DCHAINK)
template <bool D, bool GZ> struct CopyPrefetchChain<RigidBody<double, WRENCH_SATELLITE, true, D, GZ>, double> { static constexpr bool value = true; static constexpr int PACE = 0; };
template <bool D, bool GZ> struct CopyPrefetchChain<RigidBody<double, WRENCH_QUADROTOR, true, D, GZ>, double> { static constexpr bool value = true; static constexpr int PACE = 3; };
template <class T, bool D, bool GZ, int RULE> struct UnrollPolicy<RigidBody<T, WRENCH_QUADROTOR, true, D, GZ>, RULE> { static constexpr bool value = true; using Plan = QuadBodyPlan; static constexpr int NC = 4; };

// error-state columns of a rigid body: r 0-2, phi 3-5, v 6-8, w 9-11, then the controls
// Satellite (round 3, searched with tune/tune_errexp.hip: 25.3 -> 23.8 us in the same-box A/B; the round-2 plan r, v, tau x 3 | (phi, w, uF) x 3 spilled 64
// bytes per lane on the value wave): (phi0 w0 tau0) (phi1 w1 tau1) (r v w2 uF0 + f(z)) (phi2 tau2 uF1 uF2); the early part of a first tile is r, phi.
// Quadrotor: the round-2 plan stays (23.5 us; the satellite's shape costs 26.7 there -- every rotor column is as heavy as a torque column).
using SatErrPlan = TablePlanS<18, 4, 2, 6, /*r*/ 2, 2, 2, /*phi*/ 0, 1, 3, /*v*/ 2, 2, 2, /*w*/ 0, 1, 2, /*uF*/ 2, 3, 3, /*tau*/ 0, 1, 3>;
using QuadErrPlan = TablePlanS<16, 4, 2, 9, /*r*/ 3, 3, 3, /*phi*/ 0, 1, 2, /*v*/ 3, 3, 3, /*w*/ 0, 1, 2, /*u*/ 3, 3, 0, 1>;
template <class T, bool B, bool D, bool GZ, int RULE> struct ErrPlanOf<RigidBody<T, WRENCH_SATELLITE, B, D, GZ>, RULE> { using Plan = SatErrPlan; };
template <class T, bool B, bool D, bool GZ, int RULE> struct ErrPlanOf<RigidBody<T, WRENCH_QUADROTOR, B, D, GZ>, RULE> { using Plan = QuadErrPlan; };

// FiniteDifference along the cascade: a column costs the blocks it can change -- r, v, uF: the r,v block (~141 fp64 instructions of an RK4
// step); q: q + r,v (~231); w, tau and the quadrotor's rotors: everything (~312).  Four waves of ~900-1090, the early part (r, q, v) spread evenly.
using SatFdPlan = TablePlanS<19, 4, 3, 10, /*r*/ 2, 2, 2, /*q*/ 0, 0, 1, 1, /*v*/ 3, 3, 3, /*w*/ 0, 1, 2, /*uF*/ 2, 2, 3, /*tau*/ 0, 1, 3>;
using QuadFdPlan = TablePlanS<17, 4, 2, 10, /*r*/ 2, 2, 2, /*q*/ 0, 0, 1, 1, /*v*/ 3, 3, 3, /*w*/ 0, 1, 2, /*u*/ 0, 1, 3, 3>;
template <class T, bool B, bool D, bool GZ> struct FdPlanOf<RigidBody<T, WRENCH_SATELLITE, B, D, GZ>> { static constexpr bool value = true; using Plan = SatFdPlan; };
template <class T, bool B, bool D, bool GZ> struct FdPlanOf<RigidBody<T, WRENCH_QUADROTOR, B, D, GZ>> { static constexpr bool value = true; using Plan = QuadFdPlan; };

// full inertia (DIAG = false; also what a body-frame quadrotor with a gravity vector off the z axis runs on): the lean feature set (dispatch.hpp)
template <class T, int W, bool B, bool GZ> struct LeanVariant<RigidBody<T, W, B, false, GZ>> { static constexpr bool value = true; };

#define RD_RIGID_VARIANT_DECL(name)                                                                       \
    int name##_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);                     \
    int name##_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a);                   \
    int name##_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);                       \
    int name##_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a);         \
    int name##_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);                    \
    int name##_rolljac(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a);
#define RD_RIGID_VARIANT_DEFINE(name, WRENCH, BODY, DIAG) RD_RIGID_VARIANT_DEFINE_GZ(name, WRENCH, BODY, DIAG, false)
#define RD_RIGID_VARIANT_DEFINE_GZ(name, WRENCH, BODY, DIAG, GZ)                                          \
    template <class T> using name##_M = RigidBody<T, WRENCH, BODY, DIAG, GZ>;                             \
    int name##_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_eval<name##_M>(md, rule, dtype, a); } \
    int name##_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return dispatch_jacobian<name##_M>(md, rule, diff, dtype, a); } \
    int name##_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_rollout<name##_M>(md, rule, dtype, a); } \
    int name##_rolljac(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return dispatch_rolljac<name##_M>(md, rule, diff, dtype, a); } \
    int name##_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_errexp<name##_M>(md, rule, dtype, a); } \
    int name##_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return dispatch_implicit<name##_M>(md, mode, dtype, a); }
RD_RIGID_VARIANT_DECL(sat_wd) RD_RIGID_VARIANT_DECL(sat_wf) RD_RIGID_VARIANT_DECL(sat_bd) RD_RIGID_VARIANT_DECL(sat_bf)
RD_RIGID_VARIANT_DECL(quad_wd) RD_RIGID_VARIANT_DECL(quad_wf) RD_RIGID_VARIANT_DECL(quad_bd) RD_RIGID_VARIANT_DECL(quad_bf)
inline bool diag_inertia(const ModelDesc& md) { return host_is_diag3(md.params + 1); }
}  // namespace rd
