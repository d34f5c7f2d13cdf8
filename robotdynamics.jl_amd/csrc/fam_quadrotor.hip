// fam_quadrotor.hip -- RigidBody{QuatRotation} with the quadrotor wrench (test/quadrotor.jl): frame / inertia-structure selection.
#include "fam_rigid.hpp"
namespace rd {
#define RD_PICK(fn, ...) (diag_inertia(md) ? (md.frame == RDHIP_FRAME_BODY ? quad_bd_##fn(__VA_ARGS__) : quad_wd_##fn(__VA_ARGS__)) \
                                           : (md.frame == RDHIP_FRAME_BODY ? quad_bf_##fn(__VA_ARGS__) : quad_wf_##fn(__VA_ARGS__)))
int fam_quadrotor_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(eval, md, rule, dtype, a); }
int fam_quadrotor_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return RD_PICK(jacobian, md, rule, diff, dtype, a); }
int fam_quadrotor_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(rollout, md, rule, dtype, a); }
int fam_quadrotor_rolljac(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return RD_PICK(rolljac, md, rule, diff, dtype, a); }
int fam_quadrotor_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(errexp, md, rule, dtype, a); }
int fam_quadrotor_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return RD_PICK(implicit, md, mode, dtype, a); }
}  // namespace rd
