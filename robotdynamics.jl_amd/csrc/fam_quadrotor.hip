// fam_quadrotor.hip -- RigidBody{QuatRotation} with the quadrotor wrench (test/quadrotor.jl).
#include "dispatch.hpp"
namespace rd {
template <class T> using QuadWorld = RigidBody<T, WRENCH_QUADROTOR, false>;
template <class T> using QuadBody = RigidBody<T, WRENCH_QUADROTOR, true>;
template <class T, bool B> struct UnrollPolicy<RigidBody<T, WRENCH_QUADROTOR, B>, RULE_RK4> { static constexpr bool value = true; using Plan = ContiguousPlan<13 + RigidBody<T, WRENCH_QUADROTOR, B>::NU, 4>; static constexpr int NC = 4; };
int fam_quadrotor_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    return md.frame == RDHIP_FRAME_BODY ? dispatch_eval<QuadBody>(md, rule, dtype, a) : dispatch_eval<QuadWorld>(md, rule, dtype, a);
}
int fam_quadrotor_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) {
    return md.frame == RDHIP_FRAME_BODY ? dispatch_jacobian<QuadBody>(md, rule, diff, dtype, a) : dispatch_jacobian<QuadWorld>(md, rule, diff, dtype, a);
}
int fam_quadrotor_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    return md.frame == RDHIP_FRAME_BODY ? dispatch_rollout<QuadBody>(md, rule, dtype, a) : dispatch_rollout<QuadWorld>(md, rule, dtype, a);
}
}  // namespace rd
