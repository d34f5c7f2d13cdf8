// tune_ng.hip -- developer tool: more than four compute waves per tile.  The body-frame rigid bodies are bound by their compute waves (tune_stamp: 4.2 / 5.3 us of
// columns per tile against ~4 us of store issue); a fifth or sixth compute wave shares a SIMD with another one.  Back-to-back launches, outputs rotated.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
#if defined(QUAD)
using Mdl = RigidBody<double, WRENCH_QUADROTOR, true, true, true>;
constexpr int NZc = 17;
// columns: r0 r1 r2 | q0 q1 q2 q3 | v0 v1 v2 | w0 w1 w2 | u0 u1 u2 u3
using P4 = QuadBodyPlan;                                                                           // shipped: 4 + 4
using P5 = TablePlanS<17, 5, 3, 10, 0, 0, 0, 0, 1, 2, 3, 0, 4, 4, 4, 1, 2, 1, 2, 3, 3>;            // 5 + 3: waves 0 and 4 (one SIMD) light
using P5b = TablePlanS<17, 5, 3, 10, 0, 0, 0, 0, 1, 2, 3, 4, 4, 4, 0, 1, 2, 4, 1, 2, 3>;           // 5 + 3: (r q0 w0) (q1 w1 u1) (q2 w2 u2) (q3 u3 val) (v u0)
using P6 = TablePlanS<17, 6, 5, 10, 2, 2, 2, 0, 1, 2, 3, 4, 4, 5, 4, 5, 2, 3, 3, 0, 1>;            // 6 + 2
#else
using Mdl = RigidBody<double, WRENCH_SATELLITE, true, true>;
constexpr int NZc = 19;
// columns: r0 r1 r2 | q0 q1 q2 q3 | v0 v1 v2 | w0 w1 w2 | uF0 uF1 uF2 | tau0 tau1 tau2
using P4 = SatPlan;
using P5 = TablePlanS<19, 5, 3, 10, 0, 0, 0, 0, 1, 2, 3, 0, 4, 4, 4, 1, 2, 0, 4, 4, 1, 2, 3>;      // (r q0 v0 uF0) (q1 w1 tau0) (q2 w2 tau1) (q3 tau2 val) (v1 v2 w0 uF1 uF2)
using P5b = TablePlanS<19, 5, 3, 10, 0, 0, 0, 0, 1, 2, 3, 4, 4, 4, 0, 1, 2, 4, 4, 4, 1, 2, 3>;     // (r q0 w0) (q1 w1 tau0) (q2 w2 tau1) (q3 tau2 val) (v uF)
using P6 = TablePlanS<19, 6, 5, 10, 2, 2, 2, 0, 1, 2, 3, 4, 4, 5, 4, 5, 2, 0, 1, 5, 3, 3, 2>;      // 6 + 2
#endif
using Fn = Fun<Mdl, RULE_RK4>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
template <class Plan, int NC> double run(const char* name, const Fn& fn, long long N, const double* Z, const double* t, const double* dt, double** J, double** Y, hipStream_t s, int reps) {
    constexpr size_t lds = SplitLds<Fn, double, Plan, NC>::BYTES;
    auto k = jac_ad_split_kernel<Fn, double, Plan, NC>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int i) { hipLaunchKernelGGL(k, dim3(256), dim3((Plan::NG + NC) * 64), lds, s, fn, N, Z, t, dt, J[i % 4], Y[i % 4], (unsigned*)nullptr); };
    for (int i = 0; i < 2000; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    double sum = 0;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms * 1e3 / reps;
    }
    std::vector<double> h(13 * NZc * 8);
    CK(hipMemcpy(h.data(), J[0] + 13 * NZc * 5000, h.size() * 8, hipMemcpyDeviceToHost));
    double cs = 0; for (size_t i = 0; i < h.size(); ++i) cs += h[i] * (1.0 + 1e-3 * (double)(i % 97));
    printf("%-14s %8.2f us  frac %.3f  checksum %.12e\n", name, sum / 4, N * (NZc * 8 + 16.0 + 13.0 * (NZc + 1) * 8) / (sum / 4 * 1e-6) / 8e12, cs);
    return sum / 4;
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 400;
    const long long N = 65536;
    std::vector<double> hz((size_t)N * NZc), ht(N), hdt(N, 0.1);
    for (long long k = 0; k < N; ++k) { for (int i = 0; i < NZc; ++i) hz[k * NZc + i] = 0.5 + 0.8 * ((k * NZc + i) * 2654435761u % 1000) / 1000.0; ht[k] = 0.1 * k; }
    double *Z, *J[4], *Y[4], *t, *dt;
    CK(hipMalloc(&Z, N * NZc * 8)); CK(hipMalloc(&t, N * 8)); CK(hipMalloc(&dt, N * 8));
    for (int b = 0; b < 4; ++b) { CK(hipMalloc(&J[b], N * 13 * NZc * 8)); CK(hipMalloc(&Y[b], N * 13 * 8)); }
    CK(hipMemcpy(Z, hz.data(), hz.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), hdt.size() * 8, hipMemcpyHostToDevice));
    double prm[16] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, -9.81, 0.175, 1.0, 0.0245};
    Fn fn{Mdl::make(prm)};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        run<P4, 4>("4+4 shipped", fn, N, Z, t, dt, J, Y, s, reps);
        run<P5, 3>("5+3 a", fn, N, Z, t, dt, J, Y, s, reps);
        run<P5b, 3>("5+3 b", fn, N, Z, t, dt, J, Y, s, reps);
        run<P6, 2>("6+2", fn, N, Z, t, dt, J, Y, s, reps);
    }
    return 0;
}
