#include "fam_rigid.hpp"
namespace rd {
using M = RigidBody<double, WRENCH_SATELLITE, false, true>;
using Fn = Fun<M, RULE_RK4>;
template __global__ void jac_split_multi_kernel<Fn, double, SatPlan, 4, AdUnrolledBody<Fn, double, SatPlan>, 1>(Fn, MultiBatch<double>, unsigned*);
template __global__ void jac_ad_split_kernel<Fn, double, SatPlan, 4>(Fn, long long, const double*, const double*, const double*, double*, double*, unsigned*);
}
