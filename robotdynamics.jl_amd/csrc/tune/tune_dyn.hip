// tune_dyn.hip -- developer tool: static round-robin tiles against "first three tiles static, the rest from a device-wide counter" (the kernels' last argument), same binary,
// alternating passes; outputs compared.  KIND as in tune_stamp.hip (0 flagship, 1 IFT).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
#ifndef KIND
#define KIND 0
#endif
#ifndef TT
#define TT double
#endif
using Mdl = RigidBody<TT, WRENCH_SATELLITE, false, true>;
#if KIND == 1
using Fn = IftFun<Mdl>;
using Plan = typename IftPlanOf<Mdl>::Plan;
#else
using Fn = Fun<Mdl, RULE_RK4>;
using Plan = typename UnrollPolicy<Mdl, RULE_RK4>::Plan;
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 65536;
    const int reps = argc > 2 ? atoi(argv[2]) : 400;
    constexpr int NZ = 19, OW = 13 * NZ;
    std::vector<TT> hz((size_t)N * NZ); std::vector<double> ht(N), hdt(N, 0.1);
    for (long long k = 0; k < N; ++k) {
        for (int i = 0; i < NZ; ++i) hz[k * NZ + i] = (TT)(0.1 + 0.8 * ((k * NZ + i) * 2654435761u % 1000) / 1000.0);
        double q2 = 0; for (int i = 3; i < 7; ++i) q2 += hz[k * NZ + i] * hz[k * NZ + i];
        for (int i = 3; i < 7; ++i) hz[k * NZ + i] /= (TT)std::sqrt(q2);
        ht[k] = 0.1 * k;
    }
    const int NB = N <= 262144 ? 4 : 2;
    TT *Z, *J[4], *Y[4]; double *t, *dt; int* flag; unsigned* sc; constexpr size_t TS = sizeof(TT);
    CK(hipMalloc(&Z, N * NZ * TS)); CK(hipMalloc(&t, N * 8)); CK(hipMalloc(&dt, N * 8)); CK(hipMalloc(&flag, (2 + N / 64) * 4)); CK(hipMemset(flag, 0, (2 + N / 64) * 4));
    CK(hipMalloc(&sc, 64)); CK(hipMemset(sc, 0, 64));
    for (int b = 0; b < NB; ++b) { CK(hipMalloc(&J[b], N * OW * TS)); CK(hipMalloc(&Y[b], N * 13 * TS)); }
    CK(hipMemcpy(Z, hz.data(), hz.size() * TS, hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), hdt.size() * 8, hipMemcpyHostToDevice));
    double prm[10] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
#if KIND == 1
    Fn fn{Mdl::make(prm), 1e-12, flag, 10, 2, nullptr};
    auto k = implicit_ift_split_kernel<Fn, TT, Plan, 4>;
#else
    Fn fn{Mdl::make(prm)};
    auto k = jac_ad_split_kernel<Fn, TT, Plan, 4>;
    auto kd = jac_ad_split_kernel<Fn, TT, Plan, 4, true>;          // the instantiation with the dynamic schedule
#endif
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    constexpr size_t lds = SplitLds<Fn, TT, Plan, 4>::BYTES;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ngrid = sizeof(TT) == 4 ? 512 : 256;     // (fp32: two workgroups per CU)
    const int grid = (int)((N / 64) < ngrid ? (N / 64) : ngrid);
    unsigned* cur = nullptr;                           // the launch's counter words (NULL: static schedule)
#if KIND == 1
    auto kd = k;
#endif
    CK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&](int i) { if (cur) hipLaunchKernelGGL(kd, dim3(grid), dim3(512), lds, s, fn, N, Z, t, dt, J[i % NB], Y[i % NB], cur);
                               else hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, fn, N, Z, t, dt, J[i % NB], Y[i % NB], cur); };
    for (int i = 0; i < 500; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    float ms;
    for (int pass = 0; pass < 6; ++pass) {
        unsigned* v = (pass & 1) ? sc : nullptr;
        cur = v;
        for (int i = 0; i < 50; ++i) launch(i);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("KIND %d  N %lld  %s: %.2f us per launch\n", KIND, N, v ? "first 3 static + counter (>= 8 tiles per workgroup)" : "static                                              ", ms * 1e3 / reps);
    }
    CK(hipMemset(Y[0], 0, N * 13 * TS)); CK(hipMemset(Y[1], 0, N * 13 * TS)); CK(hipMemset(J[0], 0, N * OW * TS)); CK(hipMemset(J[1], 0, N * OW * TS));
    CK(hipDeviceSynchronize());                        // (the stream is non-blocking: the memsets of the null stream must be over before the launches)
    cur = nullptr; launch(0); CK(hipStreamSynchronize(s));
    cur = sc; launch(1); launch(1); CK(hipStreamSynchronize(s));
    std::vector<TT> ja((size_t)N * OW), jb((size_t)N * OW), a((size_t)N * 13), b((size_t)N * 13);
    CK(hipMemcpy(ja.data(), J[0], ja.size() * TS, hipMemcpyDeviceToHost)); CK(hipMemcpy(jb.data(), J[1], jb.size() * TS, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a.data(), Y[0], a.size() * TS, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), Y[1], b.size() * TS, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < ja.size(); ++i) bad += ja[i] != jb[i]; for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
    unsigned hc[10]; CK(hipMemcpy(hc, sc, 40, hipMemcpyDeviceToHost));
    unsigned left = 0; for (int i = 0; i < 10; ++i) left |= hc[i];
    printf("counter vs static outputs: %zu values differ; counters after the launches: %s\n", bad, left ? "NOT ZERO" : "all zero");
    return 0;
}
