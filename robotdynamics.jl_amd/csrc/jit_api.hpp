// jit_api.hpp -- entry points of the hipRTC plug-in (jit.hip) beyond those common.hpp declares: user-defined DISCRETE models and user-written
// Jacobians (user_model.hpp).  Included by api.hip and jit.hip only.
#pragma once
namespace rd {
void* jit_model_create_discrete(int n, int m, const char* step_body, const char* jac_body, const char* preamble, const double* params, int np, int quat_rotations, const int* P, int nP);
int jit_set_user_jacobian(void* h, const char* jac_body);
int jit_is_discrete(void* h);
int jit_has_user_jacobian(void* h);
}  // namespace rd
