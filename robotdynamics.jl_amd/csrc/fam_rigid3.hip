// fam_rigid3.hip -- RigidBody{MRP} and RigidBody{RodriguesParam} (n = 12; src/rigidbody.jl:80-94, :213-236; test/rigidbody_test.jl:79-98) with the
// satellite and quadrotor wrenches: dynamics, the explicit rules, ForwardAD / FiniteDifference Jacobians, rollout, ImplicitMidpoint.  Rotation kind
// and velocity frame are run-time members of the model (models.hpp RigidBody3), so one instantiation per wrench and dtype serves them all.
#include "dispatch.hpp"
namespace rd {
template <int WRENCH, class F64, class F32> static int with_model(const ModelDesc& md, int dtype, F64 f64, F32 f32) {
    const int body = md.frame == RDHIP_FRAME_BODY ? 1 : 0;
    if (dtype == RDHIP_F64) return f64(RigidBody3<double, WRENCH>::make(md.params, md.rot, body));
    if (dtype == RDHIP_F32) return f32(RigidBody3<float, WRENCH>::make(md.params, md.rot, body));
    return RDHIP_ERR_INVALID_ARG;
}
#define RD_R3(WRENCH, expr) with_model<WRENCH>(md, dtype, [&](const RigidBody3<double, WRENCH>& m) { return expr; }, [&](const RigidBody3<float, WRENCH>& m) { return expr; })
#define RD_R3_PICK(expr) (md.kind == RDHIP_MODEL_SATELLITE ? RD_R3(WRENCH_SATELLITE, expr) : RD_R3(WRENCH_QUADROTOR, expr))
int fam_rigid3_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_R3_PICK(dispatch_eval_rule(m, rule, a)); }
int fam_rigid3_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return RD_R3_PICK(dispatch_jacobian_rule(m, rule, diff, a)); }
int fam_rigid3_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_R3_PICK(dispatch_rollout_rule(m, rule, a)); }
int fam_rigid3_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return RD_R3_PICK(launch_implicit(m, mode, a)); }
}  // namespace rd
