// fam_quad_bd.hip -- RigidBody<T, WRENCH_QUADROTOR, body-frame velocity = true, diagonal inertia = true, gravity along z = true>: all rules, diff
// methods and dtypes (test/quadrotor.jl:36-46: Diagonal inertia, gravity [0, 0, -9.81]).  A body-frame quadrotor with another gravity direction takes
// the full-inertia variant (fam_quad_bf.hip), which holds both a general inertia and a general gravity vector.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE_GZ(quad_bd, WRENCH_QUADROTOR, true, true, true)
}  // namespace rd
