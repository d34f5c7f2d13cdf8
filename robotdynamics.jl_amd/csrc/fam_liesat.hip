// fam_liesat.hip -- the `Satellite` LieGroupModel of test/liemodel.jl (x = [w; q]): every operator, all rules, diff methods and dtypes.
#include "fam_lie.hpp"
namespace rd {
int liesat_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_eval<LieSatellite>(md, rule, dtype, a); }
int liesat_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return dispatch_jacobian<LieSatellite>(md, rule, diff, dtype, a); }
int liesat_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_rollout<LieSatellite>(md, rule, dtype, a); }
int liesat_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_errexp<LieSatellite>(md, rule, dtype, a); }
int liesat_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return dispatch_implicit<LieSatellite>(md, mode, dtype, a); }
}  // namespace rd
