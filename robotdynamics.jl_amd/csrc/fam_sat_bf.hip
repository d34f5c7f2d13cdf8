// fam_sat_bf.hip -- RigidBody<T, WRENCH_SATELLITE, body-frame velocity = true, diagonal inertia = false>: all rules, diff methods and dtypes.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE(sat_bf, WRENCH_SATELLITE, true, false)
}  // namespace rd
