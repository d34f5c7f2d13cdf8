// tune_c5.hip -- developer tool: the satellite half of BASELINE config 5 as ONE GPU sees the whole workload (4 096 trajectories x 256 knots, fp32, time-major):
// the fused rollout -> Jacobians kernel with a varying number of Jacobian workgroups, the rollout alone, the Jacobians alone; with -DRD_FUSED_STAMP the
// start / end times of every workgroup's role (when does the rollout finish inside the fused launch?).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize -DRD_FUSED_STAMP tune/tune_c5.hip -o tune/tune_c5
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
#ifndef TT
#define TT float
#endif
#ifndef FR                  // ring depth of the rollout role (cascade.hpp): 4 = 98 KB of LDS in fp32, 2 = 73 KB (two workgroups per CU)
#define FR 4
#endif
#ifndef FNC                 // copy waves of the fused kernel: 4 = 512 threads, one workgroup per CU; 2 (fp32) = 384 threads, two per CU
#define FNC 4
#endif
using Mdl = RigidBody<TT, WRENCH_SATELLITE, false, true>;
using Fn = Fun<Mdl, RULE_RK4>;
using Cas = Cascade<Mdl>;
// the Jacobian role of the fused kernel alone (every workgroup a consumer), for launches BEHIND a publishing rollout: what do the waits and the coherent loads cost by themselves?
struct TilesNoCoh : RolloutTilesTM<TT> { static constexpr bool COHERENT = false; };
struct TilesNoPoll : RolloutTilesTM<TT> { static constexpr bool COHERENT = false; RD_DEV void wait(const TileRef<TT>&, FusedSeen&) const {} };
template <int MW>
__global__ void __launch_bounds__(FUSED_THREADS, MW) shell_plain(Fn fn, long long N, const TT* Z, const double* tt, const double* dtt, TT* J) {
    jac_split_loop<Fn, TT, SatPlan, 4, AdUnrolledBody<Fn, TT, SatPlan>>(fn, SingleBatch<TT>{N, Z, tt, dtt, J, nullptr});
}
template <class BT>
__global__ void __launch_bounds__(FUSED_THREADS, (sizeof(TT) == 4 ? Tune::FUSED_F32_MINW : 1)) cons_only_v(Fn fn, long long K, long long M, int G, TT* Z, const double* tt, const double* dtt, TT* J, unsigned long long* progress, unsigned long long pbase) {
    const BT rt{{K, M, G, Z, tt, dtt, J, progress, pbase, FusedWalk::make(blockIdx.x, gridDim.x, 0u, 0, K * G)}};
    jac_split_loop<Fn, TT, SatPlan, 4, AdUnrolledBody<Fn, TT, SatPlan>>(fn, rt);
}
// the same over a TRAJECTORY-major stack: gathered tiles of 4 trajectories x 16 knots (RolloutTilesTJ)
template <int MW> __global__ void __launch_bounds__(FUSED_THREADS, MW) cons_only_tj(Fn fn, long long K, long long M, TT* Z, const double* tt, const double* dtt, TT* J, unsigned long long* progress, unsigned long long pbase) {
    const int Q = (int)((M + GATHER_NRUN - 1) / GATHER_NRUN);
    const RolloutTilesTJ<TT> rt{K, M, Q, Z, tt, dtt, J, progress, pbase, FusedWalk::make(blockIdx.x, gridDim.x, 0u, 0, ((K + GATHER_RUN - 1) / GATHER_RUN) * Q)};
    jac_split_loop<Fn, TT, SatPlan, 4, AdUnrolledBody<Fn, TT, SatPlan>>(fn, rt);
}
template <int MW> __global__ void __launch_bounds__(FUSED_THREADS, MW) cons_only_w(Fn fn, long long K, long long M, int G, TT* Z, const double* tt, const double* dtt, TT* J, unsigned long long* progress, unsigned long long pbase) {
    const RolloutTilesTM<TT> rt{K, M, G, Z, tt, dtt, J, progress, pbase, FusedWalk::make(blockIdx.x, gridDim.x, 0u, 0, K * G)};
    jac_split_loop<Fn, TT, SatPlan, 4, AdUnrolledBody<Fn, TT, SatPlan>>(fn, rt);
}
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
    unsigned long long* stamps; CK(hipMalloc(&stamps, 4096 * 8)); CK(hipMemset(stamps, 0, 4096 * 8));
#if defined(RD_FUSED_STAMP)
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_fused_stamp), &stamps, sizeof stamps));
#endif
    CK(hipMemcpy(Z, hz.data(), hz.size() * sizeof(TT), hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), N * 8, hipMemcpyHostToDevice));
    double prm[16] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    Fn fn{Mdl::make(prm)};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int G = (int)((M + 63) / 64);
        using FL = FusedLds<Fn, TT, Cas, FR, SatPlan, FNC>;
    const int GW = G;
    constexpr int FT = FUSED_THREADS;
        auto kf = rollout_jac_fused_kernel<Fn, TT, Cas, FR, SatPlan, FNC, true>;
    CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
    auto kj = jac_ad_split_kernel<Fn, TT, SatPlan, 4>;
    constexpr size_t ldsj = SplitLds<Fn, TT, SatPlan, 4>::BYTES;
    CK(hipFuncSetAttribute((const void*)kj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsj));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kf, FT, FL::BYTES));
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
    const bool quick = argc > 3 && argv[3][0] == 'f';
    std::vector<double> fracs = {0.0, 0.4, 0.45, 0.5, 0.55, 0.6, 0.7};
    if (quick) { fracs.clear(); for (int i = 4; i < argc; ++i) fracs.push_back(atof(argv[i])); }
    if (quick) {      // ~200 ms of load first: the clocks settle over ~100 ms, a short warm-up measures the ramp
        for (int i = 0; i < 6000; ++i) { ++epoch; hipLaunchKernelGGL(kf, dim3(256), dim3(FT), FL::BYTES, s, fn, K, M, G, 1LL, M, Z, t, dt, (const TT*)nullptr, J, prog, epoch << 32, 0LL); }
        CK(hipStreamSynchronize(s));
    }
    if (!quick) {
    timeit("rollout alone (fused kernel, grid = G)", [&]() { hipLaunchKernelGGL(kf, dim3(GW), dim3(FT), FL::BYTES, s, fn, K, M, G, 1LL, M, Z, t, dt, (const TT*)nullptr, (TT*)nullptr, (unsigned long long*)nullptr, 0ULL, 0LL); });
    timeit("Jacobians alone (256 workgroups)", [&]() { hipLaunchKernelGGL(kj, dim3(256), dim3(512), ldsj, s, fn, N, Z, t, dt, J, (TT*)nullptr, (unsigned*)nullptr); });
    {   // the non-split tile loop: 4 waves that compute and then copy, 256 threads, two workgroups per CU (a 6-wave workgroup above 128 registers never
        // shares a CU with a second one, whatever the occupancy API says: tune/occ_place.hip)
        auto ku = jac_ad_unrolled_kernel<Fn, TT, SatPlan>;
        constexpr size_t ldsu = JacLds<Fn, TT, SatPlan>::BYTES;
        CK(hipFuncSetAttribute((const void*)ku, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsu));
        int occu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occu, (const void*)ku, 256, ldsu));
        printf("tile-loop kernel: LDS %zu, API occupancy %d\n", ldsu, occu);
        for (int g : {256, 512, 768})
            { char nm[96]; snprintf(nm, sizeof nm, "Jacobians alone, 4-wave tile loop (%d workgroups)", g);
              timeit(nm, [&]() { hipLaunchKernelGGL(ku, dim3(g), dim3(256), ldsu, s, fn, N, Z, t, dt, J, (TT*)nullptr); }); }
    }
    {
        auto cons_only = cons_only_w<1>; auto cons_only2 = cons_only_w<2>;
        CK(hipFuncSetAttribute((const void*)cons_only, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
        CK(hipFuncSetAttribute((const void*)cons_only2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
        ++epoch; hipLaunchKernelGGL(kf, dim3(GW), dim3(FT), FL::BYTES, s, fn, K, M, G, 1LL, M, Z, t, dt, (const TT*)nullptr, (TT*)nullptr, prog, epoch << 32, 0LL);
        CK(hipStreamSynchronize(s));
        const unsigned long long pb = epoch << 32;
        timeit("consumer role alone behind the rollout (256)", [&]() { hipLaunchKernelGGL(cons_only, dim3(256), dim3(FT), FL::BYTES, s, fn, K, M, G, Z, t, dt, J, prog, pb); });
        {
            auto ctj1 = cons_only_tj<2>; auto ctj4 = cons_only_tj<4>;
            CK(hipFuncSetAttribute((const void*)ctj1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
            CK(hipFuncSetAttribute((const void*)ctj4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
            timeit("  trajectory-major gathered tiles, consumer alone (256)", [&]() { hipLaunchKernelGGL(ctj1, dim3(256), dim3(FT), FL::BYTES, s, fn, K, M, Z, t, dt, J, prog, pb); });
            if (2 * FL::BYTES <= 160 * 1024) {
                timeit("  trajectory-major, 128 registers, two per CU (512)", [&]() { hipLaunchKernelGGL(ctj4, dim3(512), dim3(FT), FL::BYTES, s, fn, K, M, Z, t, dt, J, prog, pb); });
                auto ctm4 = cons_only_w<4>;
                CK(hipFuncSetAttribute((const void*)ctm4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
                timeit("  time-major, 128 registers, two per CU (512)", [&]() { hipLaunchKernelGGL(ctm4, dim3(512), dim3(FT), FL::BYTES, s, fn, K, M, G, Z, t, dt, J, prog, pb); });
            }
        }
        CK(hipFuncSetAttribute((const void*)cons_only_v<TilesNoCoh>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
        CK(hipFuncSetAttribute((const void*)cons_only_v<TilesNoPoll>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
        CK(hipFuncSetAttribute((const void*)shell_plain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
        CK(hipFuncSetAttribute((const void*)shell_plain<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
        timeit("  plain descriptor, bounds (512, 1), fused LDS (192)", [&]() { hipLaunchKernelGGL(shell_plain<1>, dim3(192), dim3(FT), FL::BYTES, s, fn, N, Z, t, dt, J); });
        timeit("  plain descriptor, bounds (512, 2), fused LDS (192)", [&]() { hipLaunchKernelGGL(shell_plain<2>, dim3(192), dim3(FT), FL::BYTES, s, fn, N, Z, t, dt, J); });
        timeit("  plain descriptor, bounds (512, 2), own LDS (192)", [&]() { hipLaunchKernelGGL(shell_plain<2>, dim3(192), dim3(FT), ldsj, s, fn, N, Z, t, dt, J); });
        timeit("  ... no poll at all, plain loads (192)", [&]() { hipLaunchKernelGGL(cons_only_v<TilesNoPoll>, dim3(192), dim3(FT), FL::BYTES, s, fn, K, M, G, Z, t, dt, J, prog, pb); });
        timeit("  ... with plain loads (192)", [&]() { hipLaunchKernelGGL(cons_only_v<TilesNoCoh>, dim3(192), dim3(FT), FL::BYTES, s, fn, K, M, G, Z, t, dt, J, prog, pb); });
#if defined(RD_SPLIT_STAMP)
        {   // where does an iteration of the tile loop spend its time?  (slots: kernels.hpp RD_STAMP)
            long long* st; const size_t nst = (size_t)256 * 8 * 16; CK(hipMalloc(&st, nst * 8));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_split_stamp), &st, sizeof st));
            auto report = [&](const char* name, auto&& launch) {
                for (int i = 0; i < 3; ++i) launch();
                CK(hipStreamSynchronize(s)); CK(hipMemset(st, 0, nst * 8)); launch(); CK(hipStreamSynchronize(s));
                std::vector<long long> h(nst); CK(hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost));
                double a[8] = {0}; int n = 0;
                for (int b = 0; b < 192; ++b) for (int it = 2; it < 7; ++it) {
                    const long long* q = &h[((size_t)b * 8 + it) * 16]; const long long* qn = &h[((size_t)b * 8 + it + 1) * 16];
                    if (!q[0] || !qn[0]) continue;
                    long long cdone = std::max(std::max(q[4], q[5]), std::max(q[6], q[7]));
                    a[0] += (cdone - q[0]) * 0.01; a[1] += (q[2] - cdone) * 0.01; a[2] += (q[3] - q[2]) * 0.01; a[3] += (qn[0] - q[3]) * 0.01; a[4] += (qn[0] - q[0]) * 0.01;
                    a[5] += (q[9] - q[8]) * 0.01; a[6] += (q[10] - q[9]) * 0.01; a[7] += (q[4] - q[0]) * 0.01; ++n;
                }
                printf("%-40s per iteration: enter->columns done (slowest wave) %.2f (wave 0 %.2f), ->A %.2f, A->B %.2f, B->next enter %.2f, total %.2f us; copy: LDS->regs %.2f, stores issued in %.2f\n",
                       name, a[0] / n, a[7] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n);
            };
            report("stamps: plain descriptor (512,1)", [&]() { hipLaunchKernelGGL(shell_plain<1>, dim3(192), dim3(FT), FL::BYTES, s, fn, N, Z, t, dt, J); });
            report("stamps: consumer, no poll", [&]() { hipLaunchKernelGGL(cons_only_v<TilesNoPoll>, dim3(192), dim3(FT), FL::BYTES, s, fn, K, M, G, Z, t, dt, J, prog, pb); });
            report("stamps: consumer", [&]() { hipLaunchKernelGGL(cons_only, dim3(192), dim3(FT), FL::BYTES, s, fn, K, M, G, Z, t, dt, J, prog, pb); });
            { auto ctj = cons_only_tj<2>; CK(hipFuncSetAttribute((const void*)ctj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
              report("stamps: consumer, trajectory-major tiles", [&]() { hipLaunchKernelGGL(ctj, dim3(192), dim3(FT), FL::BYTES, s, fn, K, M, Z, t, dt, J, prog, pb); }); }
            st = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_split_stamp), &st, sizeof st));
        }
#endif
        timeit("consumer role alone, bounds (512, 2) (192)", [&]() { hipLaunchKernelGGL(cons_only2, dim3(192), dim3(FT), FL::BYTES, s, fn, K, M, G, Z, t, dt, J, prog, pb); });
        timeit("consumer role alone behind the rollout (192)", [&]() { hipLaunchKernelGGL(cons_only, dim3(192), dim3(FT), FL::BYTES, s, fn, K, M, G, Z, t, dt, J, prog, pb); });
    }
    timeit("Jacobians alone (192 workgroups)", [&]() { hipLaunchKernelGGL(kj, dim3(192), dim3(512), ldsj, s, fn, N, Z, t, dt, J, (TT*)nullptr, (unsigned*)nullptr); });
    }
    for (double split_frac : fracs)
    for (int nj : {occ * 256 - GW}) {
        if (nj <= 0) continue;
        char name[96]; snprintf(name, sizeof name, "fused: %d rollout + %d Jacobian workgroups, split at %.2f", GW, nj, split_frac);
        const long long split = (long long)(split_frac * K) * G;
#if defined(FUSED_TJ)    // the trajectory-major form of the same launch (gathered tiles; strides (K, 1))
        auto kfj = rollout_jac_fused_kernel<Fn, TT, Cas, FR, SatPlan, FNC, false>;
        CK(hipFuncSetAttribute((const void*)kfj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL::BYTES));
        timeit(name, [&]() { ++epoch; hipLaunchKernelGGL(kfj, dim3(GW + nj), dim3(FT), FL::BYTES, s, fn, K, M, G, K, 1LL, Z, t, dt, (const TT*)nullptr, J, prog, epoch << 32, 0LL); });
#else
        timeit(name, [&]() { ++epoch; hipLaunchKernelGGL(kf, dim3(GW + nj), dim3(FT), FL::BYTES, s, fn, K, M, G, 1LL, M, Z, t, dt, (const TT*)nullptr, J, prog, epoch << 32, split); });
#endif
#if defined(RD_FUSED_STAMP)
        std::vector<unsigned long long> hp(4096);
        CK(hipMemcpy(hp.data(), stamps, 4096 * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ULL; for (int b = 0; b < GW + nj; ++b) t0 = std::min(t0, hp[2048 + b]);
        double rs = 0, re = 0, js = 0, je = 0, jemin = 1e30, remin = 1e30;
        for (int b = 0; b < GW; ++b) { rs = std::max(rs, (hp[2048 + b] - t0) * 0.01); re = std::max(re, (hp[1024 + b] - t0) * 0.01); remin = std::min(remin, (hp[1024 + b] - t0) * 0.01); }
        for (int b = GW; b < GW + nj; ++b) { js = std::max(js, (hp[2048 + b] - t0) * 0.01); je = std::max(je, (hp[1024 + b] - t0) * 0.01); jemin = std::min(jemin, (hp[1024 + b] - t0) * 0.01); }
        double p2 = 0, p2min = 1e30; for (int b = 0; b < GW; ++b) { if (hp[3072 + b] > t0) { p2 = std::max(p2, (hp[3072 + b] - t0) * 0.01); p2min = std::min(p2min, (hp[3072 + b] - t0) * 0.01); } }
        printf("    rollout workgroups: last start %.1f us, end %.1f .. %.1f us (their second phase: %.1f .. %.1f);  Jacobian workgroups: last start %.1f us, end %.1f .. %.1f us\n", rs, remin, re, p2min, p2, js, jemin, je);
        CK(hipMemset(stamps + 3072, 0, 1024 * 8));
#endif
    }
    if (!quick)
    {   // check: the fused launch's Jacobians against the plain kernel's on the same rolled-out states
        TT* J2; CK(hipMalloc(&J2, N * NX * NZ * sizeof(TT)));
        CK(hipMemsetAsync(J, 0xff, N * NX * NZ * sizeof(TT), s));
        ++epoch; hipLaunchKernelGGL(kf, dim3(occ * 256), dim3(FT), FL::BYTES, s, fn, K, M, G, 1LL, M, Z, t, dt, (const TT*)nullptr, J, prog, epoch << 32, (long long)(0.4 * K) * G);
        hipLaunchKernelGGL(kj, dim3(256), dim3(512), ldsj, s, fn, N, Z, t, dt, J2, (TT*)nullptr, (unsigned*)nullptr);
        CK(hipStreamSynchronize(s));
        std::vector<TT> a((size_t)N * NX * NZ), b((size_t)N * NX * NZ);
        CK(hipMemcpy(a.data(), J, a.size() * sizeof(TT), hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), J2, b.size() * sizeof(TT), hipMemcpyDeviceToHost));
        double md = 0; size_t bad = 0, nan = 0;
        for (size_t i = 0; i < a.size(); ++i) { if (a[i] != a[i]) { ++nan; continue; } const double d = fabs((double)a[i] - (double)b[i]); if (d > md) md = d; if (d > 1e-3 * (1 + fabs((double)b[i]))) ++bad; }
        printf("check: fused vs plain Jacobians: max |diff| %.3g, %zu entries off, %zu NaN (of %zu)\n", md, bad, nan, a.size());
    }
    return 0;
}
