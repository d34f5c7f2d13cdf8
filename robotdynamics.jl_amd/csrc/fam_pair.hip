// fam_pair.hip -- SatellitePair (examples/satellite_pair.jl): dynamics, steps, rollout and Jacobians, all rules, diff methods and dtypes.
#include "fam_lie.hpp"
namespace rd {
int pair_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_eval<SatellitePair>(md, rule, dtype, a); }
int pair_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return dispatch_jacobian<SatellitePair>(md, rule, diff, dtype, a); }
int pair_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_rollout<SatellitePair>(md, rule, dtype, a); }
}  // namespace rd
