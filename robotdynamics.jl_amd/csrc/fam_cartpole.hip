// fam_cartpole.hip -- Cartpole (examples/cartpole.jl) kernel instantiations.
#include "dispatch.hpp"
namespace rd {
template <class T, int RULE> struct UnrollPolicy<Cartpole<T>, RULE> { static constexpr bool value = true; using Plan = ContiguousPlan<5, 4>; static constexpr int NC = 0; };
int fam_cartpole_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_eval<Cartpole>(md, rule, dtype, a); }
int fam_cartpole_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) { return dispatch_jacobian<Cartpole>(md, rule, diff, dtype, a); }
int fam_cartpole_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) { return dispatch_rollout<Cartpole>(md, rule, dtype, a); }
int fam_cartpole_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) { return dispatch_implicit<Cartpole>(md, mode, dtype, a); }
}  // namespace rd
