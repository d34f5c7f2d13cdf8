// fam_lie.hip -- LieGroupModels that are not RigidBodies: SatellitePair (examples/satellite_pair.jl) and the
// `Satellite` of test/liemodel.jl.
#include "dispatch.hpp"
namespace rd {
int fam_lie_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    return md.kind == RDHIP_MODEL_SATELLITE_PAIR ? dispatch_eval<SatellitePair>(md, rule, dtype, a) : dispatch_eval<LieSatellite>(md, rule, dtype, a);
}
int fam_lie_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) {
    return md.kind == RDHIP_MODEL_SATELLITE_PAIR ? dispatch_jacobian<SatellitePair>(md, rule, diff, dtype, a) : dispatch_jacobian<LieSatellite>(md, rule, diff, dtype, a);
}
int fam_lie_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    return md.kind == RDHIP_MODEL_SATELLITE_PAIR ? dispatch_rollout<SatellitePair>(md, rule, dtype, a) : dispatch_rollout<LieSatellite>(md, rule, dtype, a);
}
int fam_lie_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) {
    return md.kind == RDHIP_MODEL_SATELLITE_PAIR ? dispatch_implicit<SatellitePair>(md, mode, dtype, a) : dispatch_implicit<LieSatellite>(md, mode, dtype, a);
}
}  // namespace rd
