// fam_sat_wd.hip -- RigidBody<T, WRENCH_SATELLITE, body-frame velocity = false, diagonal inertia = true>: all rules, diff methods and dtypes.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE(sat_wd, WRENCH_SATELLITE, false, true)
}  // namespace rd
