// tune_c5.hip -- developer tool: the satellite half of BASELINE config 5 as ONE GPU sees the whole workload (4 096 trajectories x 256 knots, fp32, time-major):
// the fused rollout -> Jacobians kernel with a varying number of Jacobian workgroups, the rollout alone, the Jacobians alone; with -DRD_FUSED_STAMP the
// start / end times of every workgroup's role (when does the rollout finish inside the fused launch?).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize -DRD_FUSED_STAMP tune/tune_c5.hip -o tune/tune_c5
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../fam_rigid.hpp"
using namespace rd;
#ifndef TT
#define TT float
#endif
using Mdl = RigidBody<TT, WRENCH_SATELLITE, false, true>;
using Fn = Fun<Mdl, RULE_RK4>;
using Cas = Cascade<Mdl>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 4096, K = argc > 2 ? atoll(argv[2]) : 256, NZ = 19, NX = 13;
    const int reps = 10;
    const long long N = M * K;
    std::vector<TT> hz((size_t)N * NZ);
    for (long long k = 0; k < N; ++k) {
        for (int i = 0; i < NZ; ++i) hz[k * NZ + i] = (TT)(0.1 + 0.8 * ((k * NZ + i) * 2654435761u % 1000) / 1000.0);
        double nq = 0; for (int i = 3; i < 7; ++i) nq += (double)hz[k * NZ + i] * hz[k * NZ + i];
        for (int i = 3; i < 7; ++i) hz[k * NZ + i] = (TT)(hz[k * NZ + i] / sqrt(nq));
    }
    std::vector<double> ht(N), hdt(N, 0.01);
    for (long long k = 0; k < N; ++k) ht[k] = 0.01 * (k / M);
    TT *Z, *J; double *t, *dt; unsigned long long* prog;
    CK(hipMalloc(&Z, N * NZ * sizeof(TT))); CK(hipMalloc(&t, N * 8)); CK(hipMalloc(&dt, N * 8));
    CK(hipMalloc(&J, N * NX * NZ * sizeof(TT))); CK(hipMalloc(&prog, 4096 * 8)); CK(hipMemset(prog, 0, 4096 * 8));
    CK(hipMemcpy(Z, hz.data(), hz.size() * sizeof(TT), hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), N * 8, hipMemcpyHostToDevice));
    double prm[16] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    Fn fn{Mdl::make(prm)};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int G = (int)((M + 63) / 64);
    using FL = FusedLds<Fn, TT, Cas, 4, SatPlan, 4>;
    auto kf = rollout_jac_fused_kernel<Fn, TT, Cas, 4, SatPlan, 4, true>;
    CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
    auto kj = jac_ad_split_kernel<Fn, TT, SatPlan, 4>;
    constexpr size_t ldsj = SplitLds<Fn, TT, SatPlan, 4>::BYTES;
    CK(hipFuncSetAttribute((const void*)kj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsj));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kf, 512, FL::BYTES));
    printf("M=%lld K=%lld G=%d  fused LDS %zu B, occupancy %d workgroups per CU; Jacobian LDS %zu\n", M, K, G, (size_t)FL::BYTES, occ, ldsj);
    const double alg = (double)N * (NZ * sizeof(TT) + 16 + NX * NZ * sizeof(TT) + NX * sizeof(TT));
    unsigned long long epoch = 0;
    auto timeit = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.2f us  (%.3f of 8 TB/s on the judge's bytes)\n", name, ms * 1e3 / reps, alg / (ms * 1e-3 / reps) / 8e12);
    };
    timeit("rollout alone (fused kernel, grid = G)", [&]() { hipLaunchKernelGGL(kf, dim3(G), dim3(512), FL::BYTES, s, fn, K, M, G, 1LL, M, Z, t, dt, (const TT*)nullptr, (TT*)nullptr, (unsigned long long*)nullptr, 0ULL, 0LL); });
    timeit("Jacobians alone (256 workgroups)", [&]() { hipLaunchKernelGGL(kj, dim3(256), dim3(512), ldsj, s, fn, N, Z, t, dt, J, (TT*)nullptr, (unsigned*)nullptr); });
    timeit("Jacobians alone (192 workgroups)", [&]() { hipLaunchKernelGGL(kj, dim3(192), dim3(512), ldsj, s, fn, N, Z, t, dt, J, (TT*)nullptr, (unsigned*)nullptr); });
    for (double split_frac : {0.0, 0.3, 0.35, 0.4, 0.45, 0.5, 0.55})
    for (int nj : {256 - G}) {
        if (nj <= 0) continue;
        char name[96]; snprintf(name, sizeof name, "fused: %d rollout + %d Jacobian workgroups, split at %.2f", G, nj, split_frac);
        const long long split = (long long)(split_frac * K) * G;
        timeit(name, [&]() { ++epoch; hipLaunchKernelGGL(kf, dim3(G + nj), dim3(512), FL::BYTES, s, fn, K, M, G, 1LL, M, Z, t, dt, (const TT*)nullptr, J, prog, epoch << 32, split); });
#if defined(RD_FUSED_STAMP)
        std::vector<unsigned long long> hp(4096);
        CK(hipMemcpy(hp.data(), prog, 4096 * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ULL; for (int b = 0; b < G + nj; ++b) t0 = std::min(t0, hp[2048 + b]);
        double rs = 0, re = 0, js = 0, je = 0, jemin = 1e30, remin = 1e30;
        for (int b = 0; b < G; ++b) { rs = std::max(rs, (hp[2048 + b] - t0) * 0.01); re = std::max(re, (hp[1024 + b] - t0) * 0.01); remin = std::min(remin, (hp[1024 + b] - t0) * 0.01); }
        for (int b = G; b < G + nj; ++b) { js = std::max(js, (hp[2048 + b] - t0) * 0.01); je = std::max(je, (hp[1024 + b] - t0) * 0.01); jemin = std::min(jemin, (hp[1024 + b] - t0) * 0.01); }
        double p2 = 0, p2min = 1e30; for (int b = 0; b < G; ++b) { if (hp[3072 + b] > t0) { p2 = std::max(p2, (hp[3072 + b] - t0) * 0.01); p2min = std::min(p2min, (hp[3072 + b] - t0) * 0.01); } }
        printf("    rollout workgroups: last start %.1f us, end %.1f .. %.1f us (their second phase: %.1f .. %.1f);  Jacobian workgroups: last start %.1f us, end %.1f .. %.1f us\n", rs, remin, re, p2min, p2, js, jemin, je);
        CK(hipMemset(prog + 3072, 0, 1024 * 8));
#endif
    }
    return 0;
}
