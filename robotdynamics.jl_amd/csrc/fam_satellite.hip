// fam_satellite.hip -- RigidBody{QuatRotation} with the satellite wrench (examples/single_satellite.jl, test/rigidbody_test.jl `Body`):
// run-time selection of the velocity frame (src/rigidbody.jl:258) and of the diagonal-inertia specialisation.
#include "fam_rigid.hpp"
namespace rd {
#define RD_PICK(fn, ...) (diag_inertia(md) ? (md.frame == RDHIP_FRAME_BODY ? sat_bd_##fn(__VA_ARGS__) : sat_wd_##fn(__VA_ARGS__)) \
                                           : (md.frame == RDHIP_FRAME_BODY ? sat_bf_##fn(__VA_ARGS__) : sat_wf_##fn(__VA_ARGS__)))
int fam_satellite_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(eval, md, rule, dtype, a); }
int fam_satellite_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return RD_PICK(jacobian, md, rule, diff, dtype, a); }
int fam_satellite_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(rollout, md, rule, dtype, a); }
int fam_satellite_rolljac(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return RD_PICK(rolljac, md, rule, diff, dtype, a); }
int fam_satellite_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(errexp, md, rule, dtype, a); }
int fam_satellite_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return RD_PICK(implicit, md, mode, dtype, a); }
}  // namespace rd
