// tools/abi_bench.cpp -- back-to-back rdhip_jacobian calls through the C ABI from a plain C++ host (no Python, no torch): what a ccall host sees.
// build: g++ -O2 -I include tools/abi_bench.cpp -ldl -o tools/abi_bench ; run: tools/abi_bench <librdhip.so> [N]
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rdhip.h"
#define SYM(name) auto p_##name = reinterpret_cast<decltype(&name)>(dlsym(h, #name)); if (!p_##name) { printf("missing %s\n", #name); return 1; }
#define CK(x) do { int rc_ = (x); if (rc_) { printf("rc %d at line %d: %s\n", rc_, __LINE__, p_rdhip_last_error(ctx)); return 1; } } while (0)
int main(int argc, char** argv) {
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("%s\n", dlerror()); return 1; }
    const long long N = argc > 2 ? atoll(argv[2]) : 65536;
    SYM(rdhip_ctx_create) SYM(rdhip_model_create) SYM(rdhip_malloc) SYM(rdhip_memcpy_h2d) SYM(rdhip_jacobian) SYM(rdhip_timer_start) SYM(rdhip_timer_stop)
    SYM(rdhip_ctx_synchronize) SYM(rdhip_last_error)
    rdhip_ctx* ctx = nullptr; rdhip_model* sat = nullptr;
    CK(p_rdhip_ctx_create(0, nullptr, RDHIP_STREAM_OWN, &ctx));
    const double par[10] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    CK(p_rdhip_model_create(RDHIP_MODEL_SATELLITE, RDHIP_ROT_QUAT, RDHIP_FRAME_WORLD, par, 10, &sat));
    const int NZ = 19, NX = 13, NBUF = 4;
    std::vector<double> Z(N * NZ), t(N), dt(N, 0.1);
    unsigned long long st = 88172645463325252ULL;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (long long k = 0; k < N; ++k) {
        double q[4], nn = 0;
        for (int i = 0; i < 4; ++i) { double u1 = rnd() + 1e-12, u2 = rnd(); q[i] = std::sqrt(-2 * std::log(u1)) * std::cos(6.283185307179586 * u2); nn += q[i] * q[i]; }
        for (int i = 0; i < NZ; ++i) Z[k * NZ + i] = rnd();
        for (int i = 0; i < 4; ++i) Z[k * NZ + 3 + i] = q[i] / std::sqrt(nn);
        t[k] = 0.1 * k;
    }
    void *dZ, *dt_, *ddt, *dJ[NBUF], *dY[NBUF];
    CK(p_rdhip_malloc(ctx, &dZ, N * NZ * 8)); CK(p_rdhip_malloc(ctx, &dt_, N * 8)); CK(p_rdhip_malloc(ctx, &ddt, N * 8));
    for (int b = 0; b < NBUF; ++b) { CK(p_rdhip_malloc(ctx, &dJ[b], N * NX * NZ * 8)); CK(p_rdhip_malloc(ctx, &dY[b], N * NX * 8)); }
    CK(p_rdhip_memcpy_h2d(ctx, dZ, Z.data(), N * NZ * 8)); CK(p_rdhip_memcpy_h2d(ctx, dt_, t.data(), N * 8)); CK(p_rdhip_memcpy_h2d(ctx, ddt, dt.data(), N * 8));
    const struct { const char* name; int rule, diff; } cases[] = {{"rk4_ad", RDHIP_RK4, RDHIP_FORWARD_AD}, {"rk3_ad", RDHIP_RK3, RDHIP_FORWARD_AD}, {"rk4_fd", RDHIP_RK4, RDHIP_FINITE_DIFF},
                                                                  {"rk4_user", RDHIP_RK4, RDHIP_USER_DEFINED}, {"ift", RDHIP_IMPLICIT_MIDPOINT, RDHIP_FORWARD_AD}};
    for (int round = 0; round < 2; ++round)
        for (auto& c : cases) {
            for (int i = 0; i < 50; ++i) CK(p_rdhip_jacobian(ctx, sat, c.rule, c.diff, RDHIP_F64, N, dZ, (double*)dt_, (double*)ddt, dJ[i % NBUF], dY[i % NBUF], RDHIP_DEVICE));
            CK(p_rdhip_ctx_synchronize(ctx));
            const int R = 400; float ms = 0;
            CK(p_rdhip_timer_start(ctx));
            for (int i = 0; i < R; ++i) CK(p_rdhip_jacobian(ctx, sat, c.rule, c.diff, RDHIP_F64, N, dZ, (double*)dt_, (double*)ddt, dJ[i % NBUF], dY[i % NBUF], RDHIP_DEVICE));
            CK(p_rdhip_timer_stop(ctx, &ms));
            printf("%d %-9s %7.2f us per call  frac %.3f\n", round, c.name, ms * 1e3 / R, (double)N * 2248 / (ms * 1e-3 / R) / 8e12);
        }
    return 0;
}
