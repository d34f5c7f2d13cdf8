#include "dispatch.hpp"
namespace rd {
using Fn = Fun<Cartpole<float>, RULE_RK4>;
template __global__ void rollout_kernel<Fn, float>(Fn, long long, long long, long long, long long, float*, const double*, const double*, const float*);
}
