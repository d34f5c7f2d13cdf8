// tune_ift.hip -- developer tool: the one-launch ImplicitMidpoint Jacobian kernel (helper-wave Newton + IFT columns), back-to-back launches, outputs rotated.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
using Mdl = RigidBody<double, WRENCH_SATELLITE, false, true>;
using Fn = IftFun<Mdl>;
using Plan = typename IftPlanOf<Mdl>::Plan;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 400;
    const long long N = 65536;
    std::vector<double> hz((size_t)N * 19), ht(N), hdt(N, 0.1);
    for (long long k = 0; k < N; ++k) {
        for (int i = 0; i < 19; ++i) hz[k * 19 + i] = 0.1 + 0.8 * ((k * 19 + i) * 2654435761u % 1000) / 1000.0;
        double q2 = 0; for (int i = 3; i < 7; ++i) q2 += hz[k * 19 + i] * hz[k * 19 + i];
        for (int i = 3; i < 7; ++i) hz[k * 19 + i] /= std::sqrt(q2);
        ht[k] = 0.1 * k;
    }
    double *Z, *t, *dt, *J[4], *Y[4]; int* flag;
    CK(hipMalloc(&Z, N * 19 * 8)); CK(hipMalloc(&t, N * 8)); CK(hipMalloc(&dt, N * 8)); CK(hipMalloc(&flag, (2 + N / 64) * 4)); CK(hipMemset(flag, 0, (2 + N / 64) * 4));
    for (int b = 0; b < 4; ++b) { CK(hipMalloc(&J[b], N * 247 * 8)); CK(hipMalloc(&Y[b], N * 13 * 8)); }
    CK(hipMemcpy(Z, hz.data(), hz.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), hdt.size() * 8, hipMemcpyHostToDevice));
    double prm[10] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    Fn fn{Mdl::make(prm), 1e-12, flag, 10, 2, nullptr};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    constexpr size_t lds = SplitLds<Fn, double, Plan, 4>::BYTES;
    auto k = implicit_ift_split_kernel<Fn, double, Plan, 4>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int i) { hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, s, fn, N, Z, t, dt, J[i % 4], Y[i % 4], (unsigned*)nullptr); };
    for (int i = 0; i < 3000; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s %8.2f us\n", argc > 2 ? argv[2] : "", ms * 1e3 / reps);
    }
    std::vector<double> hj(247 * 4), hy(13 * 4); std::vector<int> hf(2 + N / 64);
    CK(hipMemcpy(hj.data(), J[0] + 247 * 1000, hj.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy.data(), Y[0] + 13 * 1000, hy.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hf.data(), flag, hf.size() * 4, hipMemcpyDeviceToHost));
    int nf = 0; for (size_t i = 1; i < hf.size(); ++i) nf += hf[i] != 0;
    double cs = 0; for (double v : hj) cs += v * 1.000001; for (double v : hy) cs += v;
    printf("checksum %.12f  flagged tiles %d\n", cs, nf);
    return 0;
}
