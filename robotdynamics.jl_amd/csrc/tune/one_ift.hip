#include "fam_rigid.hpp"
namespace rd {
using M = RigidBody<double, WRENCH_SATELLITE, false, true>;
using IFn = IftFun<M>;
using IPlan = typename IftPlanOf<M>::Plan;
template __global__ void implicit_ift_split_kernel<IFn, double, IPlan, 4>(IFn, long long, const double*, const double*, const double*, double*, double*, unsigned*);
template __global__ void implicit_midpoint_kernel<M, double, true>(M, long long, const double*, const double*, const double*, double*, double*, int, double, const int*, int);
}
