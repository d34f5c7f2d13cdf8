// fam_lie.hip -- LieGroupModels that are not RigidBodies: run-time selection between SatellitePair (examples/satellite_pair.jl) and the
// `Satellite` of test/liemodel.jl; the kernels are instantiated in fam_pair*.hip and fam_liesat.hip.
#include "fam_lie.hpp"
namespace rd {
#define RD_PICK(fn, ...) (md.kind == RDHIP_MODEL_SATELLITE_PAIR ? pair_##fn(__VA_ARGS__) : liesat_##fn(__VA_ARGS__))
int fam_lie_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(eval, md, rule, dtype, a); }
int fam_lie_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return RD_PICK(jacobian, md, rule, diff, dtype, a); }
int fam_lie_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(rollout, md, rule, dtype, a); }
int fam_lie_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return RD_PICK(errexp, md, rule, dtype, a); }
int fam_lie_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return RD_PICK(implicit, md, mode, dtype, a); }
}  // namespace rd
