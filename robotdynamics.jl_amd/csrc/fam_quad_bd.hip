// fam_quad_bd.hip -- RigidBody<T, WRENCH_QUADROTOR, body-frame velocity = true, diagonal inertia = true>: all rules, diff methods and dtypes.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE(quad_bd, WRENCH_QUADROTOR, true, true)
}  // namespace rd
