#include "fam_rigid.hpp"
namespace rd {
using M = RigidBody<double, WRENCH_SATELLITE, false, true>;
using Fn = Fun<M, RULE_RK4>;
template __global__ void rollout_jac_fused_kernel<Fn, double, Cascade<M>, 4, SatPlan, 4, TMFLAG>(Fn, long long, long long, int, long long, long long, double*, const double*, const double*, const double*, double*, unsigned long long*, unsigned long long, long long);
using Mf = RigidBody<float, WRENCH_SATELLITE, false, true>;
using Fnf = Fun<Mf, RULE_RK4>;
template __global__ void rollout_jac_fused_kernel<Fnf, float, Cascade<Mf>, 4, SatPlan, 4, TMFLAG>(Fnf, long long, long long, int, long long, long long, float*, const double*, const double*, const float*, float*, unsigned long long*, unsigned long long, long long);
}
