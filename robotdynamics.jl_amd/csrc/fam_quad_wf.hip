// fam_quad_wf.hip -- RigidBody<T, WRENCH_QUADROTOR, body-frame velocity = false, diagonal inertia = false>: all rules, diff methods and dtypes.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE(quad_wf, WRENCH_QUADROTOR, false, false)
}  // namespace rd
