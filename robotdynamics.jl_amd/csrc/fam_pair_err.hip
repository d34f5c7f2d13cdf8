// fam_pair_err.hip -- SatellitePair: fused error-state expansion and the ImplicitMidpoint kernels.
#include "fam_lie.hpp"
namespace rd {
int pair_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_errexp<SatellitePair>(md, rule, dtype, a); }
int pair_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return dispatch_implicit<SatellitePair>(md, mode, dtype, a); }
}  // namespace rd
