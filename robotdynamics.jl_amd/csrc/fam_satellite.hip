// fam_satellite.hip -- RigidBody{QuatRotation} with the satellite wrench (examples/single_satellite.jl,
// test/rigidbody_test.jl `Body`), world- and body-frame velocities.
#include "dispatch.hpp"
namespace rd {
template <class T> using SatWorld = RigidBody<T, WRENCH_SATELLITE, false>;
template <class T> using SatBody = RigidBody<T, WRENCH_SATELLITE, true>;
template <class T> using SatWorldD = RigidBody<T, WRENCH_SATELLITE, false, true>;     // diagonal inertia
template <class T> using SatBodyD = RigidBody<T, WRENCH_SATELLITE, true, true>;
// flagship: RK4 (and the continuous Jacobian) with compile-time tangent columns.  Columns: r 0-2, q 3-6, v 7-9, w 10-12,
// uF 13-15, tau 16-18.  Measured fp64 instruction counts per column of the RK4 tangent (diagonal J): r 0, v 2, q ~225,
// w ~343, uF ~57, tau ~257 (+ ~150-200 shared per wave) -> four waves of ~850-950 instead of 380/840/1260/1020 for
// contiguous ranges:   wave 0: r, q3-q5, v    wave 1: w10, w11    wave 2: q6, w12, uF (+ f(z))    wave 3: tau
using SatPlan = TablePlan<19, 4, 2, /*r*/ 0, 0, 0, /*q*/ 0, 0, 0, 2, /*v*/ 0, 0, 0, /*w*/ 1, 1, 2, /*uF*/ 2, 2, 2, /*tau*/ 3, 3, 3>;
template <class T, bool B, bool D> struct UnrollPolicy<RigidBody<T, WRENCH_SATELLITE, B, D>, RULE_RK4> { static constexpr bool value = true; using Plan = SatPlan; static constexpr int NC = 4; };
template <class T, bool B, bool D> struct UnrollPolicy<RigidBody<T, WRENCH_SATELLITE, B, D>, RULE_CONTINUOUS> { static constexpr bool value = true; using Plan = SatPlan; static constexpr int NC = 4; };
static bool diagJ(const ModelDesc& md) { return host_is_diag3(md.params + 1); }
int fam_satellite_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    if (diagJ(md)) return md.frame == RDHIP_FRAME_BODY ? dispatch_eval<SatBodyD>(md, rule, dtype, a) : dispatch_eval<SatWorldD>(md, rule, dtype, a);
    return md.frame == RDHIP_FRAME_BODY ? dispatch_eval<SatBody>(md, rule, dtype, a) : dispatch_eval<SatWorld>(md, rule, dtype, a);
}
int fam_satellite_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) {
    if (diagJ(md)) return md.frame == RDHIP_FRAME_BODY ? dispatch_jacobian<SatBodyD>(md, rule, diff, dtype, a) : dispatch_jacobian<SatWorldD>(md, rule, diff, dtype, a);
    return md.frame == RDHIP_FRAME_BODY ? dispatch_jacobian<SatBody>(md, rule, diff, dtype, a) : dispatch_jacobian<SatWorld>(md, rule, diff, dtype, a);
}
int fam_satellite_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    if (diagJ(md)) return md.frame == RDHIP_FRAME_BODY ? dispatch_rollout<SatBodyD>(md, rule, dtype, a) : dispatch_rollout<SatWorldD>(md, rule, dtype, a);
    return md.frame == RDHIP_FRAME_BODY ? dispatch_rollout<SatBody>(md, rule, dtype, a) : dispatch_rollout<SatWorld>(md, rule, dtype, a);
}
}  // namespace rd
