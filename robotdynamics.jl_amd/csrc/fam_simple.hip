// fam_simple.hip -- DoubleIntegrator{D} (test/double_integrator.jl) and the linear KAT model xdot = A x + B u.
#include "dispatch.hpp"
namespace rd {
template <class T> using DI1 = DoubleIntegrator<T, 1>;
template <class T> using DI2 = DoubleIntegrator<T, 2>;
template <class T> using DI3 = DoubleIntegrator<T, 3>;
template <class T> using Lin21 = Linear<T, 2, 1>;
template <class T> using Lin42 = Linear<T, 4, 2>;
template <class T> using Lin63 = Linear<T, 6, 3>;
#define RD_SIMPLE_SWITCH(CALL)                                                              \
    if (md.kind == RDHIP_MODEL_DOUBLE_INTEGRATOR) {                                         \
        if (md.m == 1) return CALL(DI1); if (md.m == 2) return CALL(DI2); if (md.m == 3) return CALL(DI3); \
    } else if (md.kind == RDHIP_MODEL_LINEAR) {                                             \
        if (md.n == 2 && md.m == 1) return CALL(Lin21); if (md.n == 4 && md.m == 2) return CALL(Lin42); \
        if (md.n == 6 && md.m == 3) return CALL(Lin63);                                     \
    }                                                                                       \
    return RDHIP_ERR_UNSUPPORTED;
int fam_simple_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
#define CALL(M) dispatch_eval<M>(md, rule, dtype, a)
    RD_SIMPLE_SWITCH(CALL)
#undef CALL
}
int fam_simple_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) {
#define CALL(M) dispatch_jacobian<M>(md, rule, diff, dtype, a)
    RD_SIMPLE_SWITCH(CALL)
#undef CALL
}
int fam_simple_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
#define CALL(M) dispatch_rollout<M>(md, rule, dtype, a)
    RD_SIMPLE_SWITCH(CALL)
#undef CALL
}
int fam_simple_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) {
#define CALL(M) dispatch_implicit<M>(md, mode, dtype, a)
    RD_SIMPLE_SWITCH(CALL)
#undef CALL
}
}  // namespace rd
