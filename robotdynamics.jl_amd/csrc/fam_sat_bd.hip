// fam_sat_bd.hip -- RigidBody<T, WRENCH_SATELLITE, body-frame velocity = true, diagonal inertia = true>: all rules, diff methods and dtypes.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE(sat_bd, WRENCH_SATELLITE, true, true)
}  // namespace rd
