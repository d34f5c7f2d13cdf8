// fam_quad_wd.hip -- RigidBody<T, WRENCH_QUADROTOR, body-frame velocity = false, diagonal inertia = true>: all rules, diff methods and dtypes.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE(quad_wd, WRENCH_QUADROTOR, false, true)
}  // namespace rd
