// tune_errexp.hip -- developer tool: the fused error-state expansion kernel under different column plans (back-to-back launches, outputs rotated).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
#ifdef QUAD
using Mdl = RigidBody<double, WRENCH_QUADROTOR, false, true>;
constexpr int NZc = 17, NEc = 16;
#else
using Mdl = RigidBody<double, WRENCH_SATELLITE, false, true>;
constexpr int NZc = 19, NEc = 18;
#endif
using Fn = ErrFun<Mdl, RULE_RK4>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
// columns: r0 r1 r2 | phi0 phi1 phi2 | v0 v1 v2 | w0 w1 w2 | uF0 uF1 uF2 | tau0 tau1 tau2
#ifdef QUAD
using P0 = TablePlanS<16, 4, 2, 9, 3, 3, 3, 0, 1, 2, 3, 3, 3, 0, 1, 2, 3, 3, 0, 1>;     // shipped
using P1 = TablePlanS<16, 4, 2, 6, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 0, 1, 2, 3>;
using P2 = TablePlanS<16, 4, 2, 6, 3, 3, 3, 0, 1, 3, 3, 3, 3, 0, 1, 2, 0, 1, 2, 3>;
using P3 = TablePlanS<16, 4, 3, 6, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 0, 1, 2, 3>;
using P4 = TablePlanS<16, 4, 2, 9, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 0, 1, 2, 3>;
using P5 = TablePlanS<16, 4, 2, 6, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 0, 1, 3, 3>;
#else
using P0 = TablePlanS<18, 4, 0, 9, 3, 3, 3, 0, 1, 2, 3, 3, 3, 0, 1, 2, 0, 1, 2, 3, 3, 3>;     // shipped in rounds 2-3
using P1 = TablePlanS<18, 4, 2, 9, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 3, 3, 3, 0, 1, 3>;     // (phi0 w0 tau0) (phi1 w1 tau1) (w2 r v +value) (phi2 tau2 uF), early 9
using P2 = TablePlanS<18, 4, 2, 6, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 3, 3, 3, 0, 1, 3>;     // early part = r, phi only (6 columns)
using P3 = TablePlanS<18, 4, 2, 6, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 2, 3, 3, 0, 1, 3>;     // + one uF to wave 2
using P4 = TablePlanS<18, 4, 2, 3, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 3, 3, 3, 0, 1, 3>;     // early part = r only
using P5 = TablePlanS<18, 4, 2, 12, 2, 2, 2, 0, 1, 3, 2, 2, 2, 0, 1, 2, 3, 3, 3, 0, 1, 3>;    // early part = r, phi, v, w (12 columns)
#endif
template <class Plan> void run(const char* name, const Fn& fn, long long N, const double* Z, const double* t, const double* dt, double** E, double** Y, hipStream_t s, int reps) {
    constexpr size_t lds = SplitLds<Fn, double, Plan, 4>::BYTES;
    auto k = jac_errexp_split_kernel<Fn, double, Plan, 4>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int i) { hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, s, fn, N, Z, t, dt, E[i % 4], Y[i % 4], (unsigned*)nullptr); };
    for (int i = 0; i < 3000; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    double best = 1e30, sum = 0;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps; sum += us; if (us < best) best = us;
    }
    printf("%-4s %8.2f us (best %8.2f)\n", name, sum / 5, best);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 400;
    const long long N = 65536;
    std::vector<double> hz((size_t)(N + 1) * NZc), ht(N + 1), hdt(N + 1, 0.1);
    for (long long k = 0; k <= N; ++k) { for (int i = 0; i < NZc; ++i) hz[k * NZc + i] = 0.5 + 0.8 * ((k * NZc + i) * 2654435761u % 1000) / 1000.0; ht[k] = 0.1 * k; }
    double *Z, *t, *dt, *E[4], *Y[4];
    CK(hipMalloc(&Z, (N + 1) * NZc * 8)); CK(hipMalloc(&t, (N + 1) * 8)); CK(hipMalloc(&dt, (N + 1) * 8));
    for (int b = 0; b < 4; ++b) { CK(hipMalloc(&E[b], N * 12 * NEc * 8)); CK(hipMalloc(&Y[b], N * 13 * 8)); }
    CK(hipMemcpy(Z, hz.data(), hz.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), hdt.size() * 8, hipMemcpyHostToDevice));
    double prm[16] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, -9.81, 0.175, 1.0, 0.0245};
    Fn fn{Mdl::make(prm)};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        run<P0>("P0", fn, N, Z, t, dt, E, Y, s, reps); run<P1>("P1", fn, N, Z, t, dt, E, Y, s, reps); run<P2>("P2", fn, N, Z, t, dt, E, Y, s, reps);
        run<P3>("P3", fn, N, Z, t, dt, E, Y, s, reps); run<P4>("P4", fn, N, Z, t, dt, E, Y, s, reps); run<P5>("P5", fn, N, Z, t, dt, E, Y, s, reps);
    }
    return 0;
}
