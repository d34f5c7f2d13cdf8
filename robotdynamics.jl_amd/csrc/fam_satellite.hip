// fam_satellite.hip -- RigidBody{QuatRotation} with the satellite wrench (examples/single_satellite.jl,
// test/rigidbody_test.jl `Body`), world- and body-frame velocities.
#include "dispatch.hpp"
namespace rd {
template <class T> using SatWorld = RigidBody<T, WRENCH_SATELLITE, false>;
template <class T> using SatBody = RigidBody<T, WRENCH_SATELLITE, true>;
// flagship: RK4 (and the continuous Jacobian) with compile-time tangent columns
template <class T, bool B> struct UnrollPolicy<RigidBody<T, WRENCH_SATELLITE, B>, RULE_RK4> { static constexpr bool value = true; using Plan = ContiguousPlan<13 + RigidBody<T, WRENCH_SATELLITE, B>::NU, 4>; static constexpr int NC = 4; };
template <class T, bool B> struct UnrollPolicy<RigidBody<T, WRENCH_SATELLITE, B>, RULE_CONTINUOUS> { static constexpr bool value = true; using Plan = ContiguousPlan<13 + RigidBody<T, WRENCH_SATELLITE, B>::NU, 4>; static constexpr int NC = 4; };
int fam_satellite_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    return md.frame == RDHIP_FRAME_BODY ? dispatch_eval<SatBody>(md, rule, dtype, a) : dispatch_eval<SatWorld>(md, rule, dtype, a);
}
int fam_satellite_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) {
    return md.frame == RDHIP_FRAME_BODY ? dispatch_jacobian<SatBody>(md, rule, diff, dtype, a) : dispatch_jacobian<SatWorld>(md, rule, diff, dtype, a);
}
int fam_satellite_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    return md.frame == RDHIP_FRAME_BODY ? dispatch_rollout<SatBody>(md, rule, dtype, a) : dispatch_rollout<SatWorld>(md, rule, dtype, a);
}
}  // namespace rd
