// tune_plan.hip -- developer tool: the flagship ForwardAD kernel under different column plans (back-to-back launches on one stream, outputs rotated).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
#ifndef TT
#define TT double
#endif
#ifndef BODYF
#define BODYF false
#endif
#ifndef DYNK                // true: the instantiation that carries the dynamic tile schedule (what the library launches for fp64, with or without counter words)
#define DYNK false
#endif
#ifndef DIAG
#define DIAG true
#endif
#ifndef RULE
#define RULE RULE_RK4
#endif
#if defined(PAIR)
using Mdl = SatellitePair<TT>;
constexpr int NZc = 20;
#elif defined(QUAD)
using Mdl = RigidBody<TT, WRENCH_QUADROTOR, BODYF, DIAG, BODYF>;
constexpr int NZc = 17;
#else
using Mdl = RigidBody<TT, WRENCH_SATELLITE, BODYF, DIAG>;
constexpr int NZc = 19;
#endif
using Fn = Fun<Mdl, RULE>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
// columns: r0 r1 r2 | q0 q1 q2 q3 | v0 v1 v2 | w0 w1 w2 | uF0 uF1 uF2 | tau0 tau1 tau2
#if defined(PAIR)
using P0 = TablePlan<20, 4, 0, 2, 2, 2, 2, 0, 0, 0, 3, 3, 3, 3, 1, 1, 1, 0, 2, 2, 1, 3, 3>;        // shipped (PairPlan)
using P1 = TablePlan<20, 4, 1, 2, 2, 2, 2, 0, 0, 0, 3, 3, 3, 3, 1, 1, 1, 0, 2, 2, 1, 3, 3>;
using P2 = TablePlan<20, 4, 2, 2, 2, 2, 2, 0, 0, 0, 3, 3, 3, 3, 1, 1, 1, 0, 2, 2, 1, 3, 3>;
using P3 = TablePlan<20, 4, 3, 2, 2, 2, 2, 0, 0, 0, 3, 3, 3, 3, 1, 1, 1, 0, 2, 2, 1, 3, 3>;
using P4 = TablePlanS<20, 4, 0, 7, 2, 2, 2, 2, 0, 0, 0, 3, 3, 3, 3, 1, 1, 1, 0, 2, 2, 1, 3, 3>;
using P5 = TablePlanS<20, 4, 2, 4, 0, 1, 2, 3, 0, 0, 2, 0, 1, 2, 3, 1, 1, 3, 0, 2, 2, 1, 3, 3>;    // one q1 column per wave first (early 4), value from wave 2
#elif defined(QUAD) && BODYF
// body frame: the v columns are no longer free (r' = R(q) v, v' = .. - w x v): spread them.  Round 6, second batch: around the shipped plan (P0 here)
using P0 = TablePlanS<17, 4, 3, 10, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1, 0, 1, 2, 3, 2, 0, 1>;     // shipped (QuadBodyPlan)
using P1 = TablePlanS<17, 4, 3, 7,  2, 2, 2, 0, 1, 2, 3, 3, 0, 1, 0, 1, 2, 3, 2, 0, 1>;     // early part = r, q only
using P2 = TablePlanS<17, 4, 3, 13, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1, 0, 1, 2, 3, 2, 0, 1>;     // early part = r, q, v, w
using P3 = TablePlanS<17, 4, 2, 10, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1, 0, 1, 2, 3, 2, 0, 1>;     // value from wave 2
using P4 = TablePlanS<17, 4, 3, 10, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1, 0, 1, 2, 2, 3, 0, 1>;     // rotors 2, 3, 0, 1
using P5 = TablePlanS<17, 4, 0, 10, 3, 3, 3, 0, 1, 2, 3, 3, 0, 1, 0, 1, 2, 2, 2, 0, 1>;     // r on 3, rotors 2, 2, 0, 1, value from 0
#elif defined(QUAD)
using P0 = TablePlanS<17, 4, 2, 10, 2, 2, 2, 0, 1, 2, 3, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1>;     // shipped (QuadPlan)
using P1 = TablePlanS<17, 4, 0, 10, 2, 2, 2, 0, 1, 2, 3, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1>;
using P2 = TablePlanS<17, 4, 1, 10, 2, 2, 2, 0, 1, 2, 3, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1>;
using P3 = TablePlanS<17, 4, 3, 10, 2, 2, 2, 0, 1, 2, 3, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1>;
using P4 = TablePlanS<17, 4, 2, 7, 2, 2, 2, 0, 1, 2, 3, 2, 2, 2, 0, 1, 2, 3, 3, 0, 1>;
using P5 = TablePlanS<17, 4, 2, 10, 3, 3, 3, 0, 1, 2, 3, 2, 2, 2, 0, 1, 2, 3, 2, 0, 1>;
#else
using P0 = TablePlanS<19, 4, 3, 10, 0, 0, 0, 0, 1, 2, 3, 1, 1, 1, 0, 1, 2, 0, 1, 1, 3, 3, 2>;     // SatPlan (value from wave 3)
#if BODYF
using P1 = TablePlanS<19, 4, 0, 10, 0, 0, 0, 0, 1, 2, 3, 0, 0, 0, 0, 1, 2, 1, 1, 1, 2, 3, 3>;     // body frame: (r q3 v w10) (q4 w11 uF) (q5 w12 tau16) (q6 tau17 tau18)
#else
using P1 = TablePlanS<19, 4, 3, 10, 0, 0, 0, 0, 1, 2, 3, 0, 1, 3, 0, 1, 2, 0, 1, 1, 3, 3, 2>;     // v spread 0, 1, 3
#endif
using P2 = TablePlanS<19, 4, 3, 10, 0, 0, 0, 0, 1, 2, 3, 0, 1, 1, 0, 1, 2, 0, 0, 1, 3, 3, 2>;     // v 0,1,1; uF 0,0,1
using P3 = TablePlanS<19, 4, 3, 10, 0, 0, 0, 0, 1, 2, 3, 0, 0, 1, 0, 1, 2, 0, 1, 1, 3, 3, 2>;     // v 0,0,1
using P4 = TablePlanS<19, 4, 2, 10, 0, 0, 0, 0, 1, 2, 3, 0, 1, 3, 0, 1, 2, 0, 1, 1, 3, 3, 2>;     // v spread, value from wave 2
using P5 = TablePlanS<19, 4, 3, 10, 1, 1, 1, 0, 1, 2, 3, 0, 1, 3, 0, 1, 2, 0, 0, 1, 3, 3, 2>;     // r on wave 1, v spread, uF 0,0,1
#endif
template <class Plan> void run(const char* name, const Fn& fn, long long N, const TT* Z, const double* t, const double* dt, TT** J, TT** Y, hipStream_t s, int reps) {
#if !defined(FDK)
    constexpr size_t lds = SplitLds<Fn, TT, Plan, 4>::BYTES;
#endif
#if defined(FDK)        // the FiniteDifference kernel along the cascade instead (the column plan is the model's own FdPlanOf: one row of output, run<P0> only)
    using FP = typename FdPlanOf<Mdl>::Plan;
    constexpr size_t lds = SplitLds<Fn, TT, FP, 4>::BYTES;
    auto k = jac_fd_cascade_kernel<Fn, TT, FP, Cascade<Mdl>, 4, DYNK>;
#elif defined(CHAINK)   // the UserDefined chain-rule kernel (the model's ForwardAD column plan)
    auto k = jac_chain_split_kernel<Fn, TT, Plan, 4, DYNK>;
#else
    auto k = jac_ad_split_kernel<Fn, TT, Plan, 4, DYNK>;
#endif
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    static const int grid = getenv("TP_GRID") ? atoi(getenv("TP_GRID")) : 256;
    auto launch = [&](int i) { hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, fn, N, Z, t, dt, J[i % 4], Y[i % 4], (unsigned*)nullptr); };
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
    printf("%-4s %8.2f us (best %8.2f)  frac %.3f\n", name, sum / 5, best, N * (NZc * sizeof(TT) + 16.0 + (double)Mdl::NX * (NZc + 1) * sizeof(TT)) / (sum / 5 * 1e-6) / 8e12);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 400;
    const long long N = getenv("TP_N") ? atoll(getenv("TP_N")) : 65536 * (8 / sizeof(TT));
    std::vector<TT> hz((size_t)N * NZc); std::vector<double> ht(N), hdt(N, 0.1);
    for (long long k = 0; k < N; ++k) { for (int i = 0; i < NZc; ++i) hz[k * NZc + i] = (TT)(0.5 + 0.8 * ((k * NZc + i) * 2654435761u % 1000) / 1000.0); ht[k] = 0.1 * k; }
    TT *Z, *J[4], *Y[4]; double *t, *dt;
    CK(hipMalloc(&Z, N * NZc * sizeof(TT))); CK(hipMalloc(&t, N * 8)); CK(hipMalloc(&dt, N * 8));
    for (int b = 0; b < 4; ++b) { CK(hipMalloc(&J[b], N * Mdl::NX * NZc * sizeof(TT))); CK(hipMalloc(&Y[b], N * Mdl::NX * sizeof(TT))); }
    CK(hipMemcpy(Z, hz.data(), hz.size() * sizeof(TT), hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), hdt.size() * 8, hipMemcpyHostToDevice));
#if defined(PAIR)
    double prm[18] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 2, 0, 0, 0, 2, 0, 0, 0, 2};
#else
    double prm[16] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, -9.81, 0.175, 1.0, 0.0245};
#endif
    Fn fn{Mdl::make(prm)};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        run<P0>("P0", fn, N, Z, t, dt, J, Y, s, reps); run<P1>("P1", fn, N, Z, t, dt, J, Y, s, reps); run<P2>("P2", fn, N, Z, t, dt, J, Y, s, reps);
        run<P3>("P3", fn, N, Z, t, dt, J, Y, s, reps); run<P4>("P4", fn, N, Z, t, dt, J, Y, s, reps); run<P5>("P5", fn, N, Z, t, dt, J, Y, s, reps);
    }
    return 0;
}
