/* c_abi_demo.c -- plain-C use of the librdhip C ABI (include/rdhip.h), host-memory calling convention (RDHIP_HOST):
 * what a Julia `ccall` or any other FFI host does.  Build:  gcc -std=c99 -Iinclude examples/c_abi_demo.c -Lrobotdynamics.jl_amd/lib -lrdhip -lm
 * Without a GPU the compute call reports RDHIP_ERR_NO_DEVICE (there is no CPU fallback); the integer queries work everywhere. */
#include <stdio.h>
#include <stdlib.h>
#include "rdhip.h"

int main(void) {
    /* RigidBody{QuatRotation} satellite: params = {mass, J[9] column-major} (examples/single_satellite.jl:33-35) */
    const double params[10] = {1.0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    rdhip_model* model = NULL;
    if (rdhip_model_create(RDHIP_MODEL_SATELLITE, RDHIP_ROT_QUAT, RDHIP_FRAME_WORLD, params, 10, &model) != RDHIP_OK) {
        fprintf(stderr, "model: %s\n", rdhip_last_error(NULL));
        return 1;
    }
    int n, m, p, e;
    rdhip_model_dims(model, &n, &m, &p, &e);
    printf("dims: n=%d m=%d p=%d errstate_dim=%d\n", n, m, p, e);          /* 13 6 13 12 (test/rigidbody_test.jl:60) */
    if (n != 13 || m != 6 || e != 12) return 2;

    int count = 0;
    rdhip_device_count(&count);
    rdhip_ctx* ctx = NULL;
    int rc = rdhip_ctx_create(0, NULL, RDHIP_STREAM_OWN, &ctx);
    if (rc != RDHIP_OK) {
        printf("no GPU context (%d devices): %s -- compute calls are unavailable, exiting cleanly\n", count, rdhip_last_error(NULL));
        rdhip_model_destroy(model);
        return rc == RDHIP_ERR_NO_DEVICE ? 0 : 3;
    }
    /* a trajectory of N knot points: z = [r q v w | u], t, dt */
    enum { N = 128 };
    double* Z = (double*)calloc((size_t)N * 19, sizeof(double));
    double* t = (double*)calloc(N, sizeof(double));
    double* dt = (double*)calloc(N, sizeof(double));
    double* J = (double*)malloc((size_t)N * 13 * 19 * sizeof(double));
    double* xn = (double*)malloc((size_t)N * 13 * sizeof(double));
    for (int k = 0; k < N; ++k) { Z[k * 19 + 3] = 1.0; Z[k * 19 + 10] = 0.1 * k; Z[k * 19 + 13] = 1.0; t[k] = 0.1 * k; dt[k] = 0.1; }
    /* jacobian!(StaticReturn(), ForwardAD(), DiscretizedDynamics{RK4}(model), J[k], xn[k], Z[k]) for all k */
    rc = rdhip_jacobian(ctx, model, RDHIP_RK4, RDHIP_FORWARD_AD, RDHIP_F64, N, Z, t, dt, J, xn, RDHIP_HOST);
    if (rc != RDHIP_OK) { fprintf(stderr, "jacobian: %s\n", rdhip_last_error(ctx)); return 4; }
    printf("d r'/d r of knot 5 (must be 1): %g   d r'/d v (must be dt = 0.1): %g   x_next[r_x] of knot 5: %g\n",
           J[5 * 247 + 0], J[5 * 247 + 7 * 13 + 0], xn[5 * 13]);
    free(Z); free(t); free(dt); free(J); free(xn);
    rdhip_ctx_destroy(ctx);
    rdhip_model_destroy(model);
    return 0;
}
