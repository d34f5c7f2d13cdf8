// fam_quadrotor.hip -- RigidBody{QuatRotation} with the quadrotor wrench (test/quadrotor.jl): frame / inertia-structure selection.
#include "fam_rigid.hpp"
namespace rd {
// params = {mass, J[9], gravity[3], ...}: the body-frame diagonal-inertia variant is compiled for gravity along z (models.hpp RigidBody GZ)
static inline bool gravity_along_z(const ModelDesc& md) { return md.params[10] == 0 && md.params[11] == 0; }
#define RD_PICK(fn, ...) (md.frame == RDHIP_FRAME_BODY ? ((diag_inertia(md) && gravity_along_z(md)) ? quad_bd_##fn(__VA_ARGS__) : quad_bf_##fn(__VA_ARGS__)) \
                                                       : (diag_inertia(md) ? quad_wd_##fn(__VA_ARGS__) : quad_wf_##fn(__VA_ARGS__)))
int fam_quadrotor_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(eval, md, rule, dtype, a); }
int fam_quadrotor_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return RD_PICK(jacobian, md, rule, diff, dtype, a); }
int fam_quadrotor_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(rollout, md, rule, dtype, a); }
int fam_quadrotor_rolljac(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return RD_PICK(rolljac, md, rule, diff, dtype, a); }
int fam_quadrotor_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(errexp, md, rule, dtype, a); }
int fam_quadrotor_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return RD_PICK(implicit, md, mode, dtype, a); }
}  // namespace rd
