// fam_lie.hpp -- shared by the LieGroupModel translation units (SatellitePair of examples/satellite_pair.jl, `Satellite` of test/liemodel.jl):
// column plans and the per-model entry points; one translation unit per model so that they compile in parallel.
#pragma once
#include "dispatch.hpp"
namespace rd {
// SatellitePair columns: q1 0-3, w1 4-6, q2 7-10, w2 11-13, tau1 14-16, tau2 17-19.  The two bodies are decoupled, so with compile-time
// tangent seeds half of every column folds away; per-column costs as for the single satellite (w > tau > q) give four waves of about
// w,w,w,tau | w,w,w,tau | tau,tau,q,q,q,q | tau,tau,q,q,q,q.
using PairPlan = TablePlan<20, 4, 0, /*q1*/ 2, 2, 2, 2, /*w1*/ 0, 0, 0, /*q2*/ 3, 3, 3, 3, /*w2*/ 1, 1, 1, /*tau1*/ 0, 2, 2, /*tau2*/ 1, 3, 3>;
template <class T, int RULE> struct UnrollPolicy<SatellitePair<T>, RULE> { static constexpr bool value = true; using Plan = PairPlan; static constexpr int NC = 4; };
// error-state columns: phi1 0-2, w1 3-5, phi2 6-8, w2 9-11, tau1 12-14, tau2 15-17 (same grouping as PairPlan); LieSatellite: w 0-2, phi 3-5, u 6-8
using PairErrPlan = TablePlan<18, 4, 0, /*phi1*/ 2, 2, 2, /*w1*/ 0, 0, 0, /*phi2*/ 3, 3, 3, /*w2*/ 1, 1, 1, /*tau1*/ 0, 2, 2, /*tau2*/ 1, 3, 3>;
template <class T, int RULE> struct ErrPlanOf<SatellitePair<T>, RULE> { using Plan = PairErrPlan; };
template <class T, int RULE> struct ErrPlanOf<LieSatellite<T>, RULE> { using Plan = ContiguousPlan<9, 4>; };
// LieSatellite columns: w 0-2, q 3-6, u 7-9.  Its ten tangents at once (the small-model kernel: Dual<T, 10> through four RK4 stages of a model that
// normalises its quaternion) took 512 registers, spilled and ran at 0.26-0.31 of the roofline (one wave per SIMD); one tangent per pass in the
// split-role tile loop like the larger rigid bodies: (w0 q0 u0) (w1 q1 u1) (w2 q2) (q3 u2 + f(z)).
using LieSatPlan = TablePlan<10, 4, 3, /*w*/ 0, 1, 2, /*q*/ 0, 1, 2, 3, /*u*/ 0, 1, 3>;
template <class T, int RULE> struct UnrollPolicy<LieSatellite<T>, RULE> { static constexpr bool value = true; using Plan = LieSatPlan; static constexpr int NC = 4; };
#define RD_LIE_MODEL_DECL(name)                                                                          \
    int name##_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);                       \
    int name##_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a);         \
    int name##_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);                    \
    int name##_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);                     \
    int name##_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a);
RD_LIE_MODEL_DECL(pair) RD_LIE_MODEL_DECL(liesat)
}  // namespace rd
