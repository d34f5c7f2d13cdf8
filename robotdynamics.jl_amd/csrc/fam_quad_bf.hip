// fam_quad_bf.hip -- RigidBody<T, WRENCH_QUADROTOR, body-frame velocity = true, diagonal inertia = false>: all rules, diff methods and dtypes.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE(quad_bf, WRENCH_QUADROTOR, true, false)
}  // namespace rd
