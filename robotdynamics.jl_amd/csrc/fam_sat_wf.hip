// fam_sat_wf.hip -- RigidBody<T, WRENCH_SATELLITE, body-frame velocity = false, diagonal inertia = false>: all rules, diff methods and dtypes.
#include "fam_rigid.hpp"
namespace rd {
RD_RIGID_VARIANT_DEFINE(sat_wf, WRENCH_SATELLITE, false, false)
}  // namespace rd
