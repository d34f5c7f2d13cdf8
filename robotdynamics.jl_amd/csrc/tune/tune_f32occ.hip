// tune_f32occ.hip -- developer tool: the fp32 satellite ForwardAD tile kernel with 4 compute + NC copy waves at different occupancies (is one 8-wave
// workgroup per CU -- one compute wave per SIMD -- what bounds the fp32 kernel?).  Build with -DRD_F32_MINB=<waves per SIMD the registers must allow>.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
using TT = float;
#ifndef NCW
#define NCW 2
#endif
using Mdl = RigidBody<TT, WRENCH_SATELLITE, false, true>;
using Fn = Fun<Mdl, RULE_RK4>;
__global__ void clk_probe(unsigned long long* out, long long ticks) {     // shader clock while other work runs: s_memtime cycles per 100 MHz wall-clock tick
    const unsigned long long t0 = wall_clock64(), c0 = clock64();
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    out[0] = clock64() - c0; out[1] = wall_clock64() - t0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const long long NZ = 19, NX = 13;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto kj = jac_ad_split_kernel<Fn, TT, SatPlan, NCW>;
    constexpr size_t ldsj = SplitLds<Fn, TT, SatPlan, NCW>::BYTES;
    constexpr int threads = (4 + NCW) * 64;
    CK(hipFuncSetAttribute((const void*)kj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsj));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kj, threads, ldsj));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)kj));
    printf("NC=%d threads=%d LDS %zu B, %d VGPRs, %zu B scratch, occupancy %d workgroups per CU\n", NCW, threads, ldsj, fa.numRegs, (size_t)fa.localSizeBytes, occ);
    double prm[16] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    Fn fn{Mdl::make(prm)};
    for (long long N : {131072LL, 1048576LL}) {
        std::vector<TT> hz((size_t)N * NZ);
        for (long long k = 0; k < N; ++k) for (int i = 0; i < NZ; ++i) hz[k * NZ + i] = (TT)(0.1 + 0.8 * ((k * NZ + i) * 2654435761u % 1000) / 1000.0);
        std::vector<double> ht(N, 0.0), hdt(N, 0.01);
        TT *Z, *J[4], *Y[4]; double *t, *dt;
        CK(hipMalloc(&Z, N * NZ * sizeof(TT))); CK(hipMalloc(&t, N * 8)); CK(hipMalloc(&dt, N * 8));
        const int nb = N > 500000 ? 2 : 4;
        for (int b = 0; b < nb; ++b) { CK(hipMalloc(&J[b], N * NX * NZ * sizeof(TT))); CK(hipMalloc(&Y[b], N * NX * sizeof(TT))); }
        CK(hipMemcpy(Z, hz.data(), hz.size() * sizeof(TT), hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), N * 8, hipMemcpyHostToDevice));
        const double alg = (double)N * (NZ * sizeof(TT) + 16 + NX * NZ * sizeof(TT) + NX * sizeof(TT));
        for (int per_cu : {1, 2, 3}) {
            if (per_cu > occ) continue;
            const int grid = 256 * per_cu, reps = N > 500000 ? 20 : 200;
            auto launch = [&](int i) { hipLaunchKernelGGL(kj, dim3(grid), dim3(threads), ldsj, s, fn, N, Z, t, dt, J[i % nb], Y[i % nb], (unsigned*)nullptr); };
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("N=%-8lld grid=%-4d  %8.2f us   %.3f of 8 TB/s", N, grid, ms * 1e3 / reps, alg / (ms * 1e-3 / reps) / 8e12);
            {   // shader clock during a run of the same launches
                static unsigned long long* dp = nullptr; static hipStream_t s2 = nullptr;
                if (!dp) { CK(hipMalloc(&dp, 16)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); }
                const double span_us = ms * 1e3 / reps * 8;
                for (int i = 0; i < 12; ++i) launch(i);
                hipLaunchKernelGGL(clk_probe, dim3(1), dim3(64), 0, s2, dp, (long long)(span_us * 100));
                CK(hipDeviceSynchronize());
                unsigned long long hp[2]; CK(hipMemcpy(hp, dp, 16, hipMemcpyDeviceToHost));
                printf("   shader clock under this load: %.0f MHz\n", (double)hp[0] / ((double)hp[1] * 0.01));
            }
        }
        CK(hipFree(Z)); CK(hipFree(t)); CK(hipFree(dt)); for (int b = 0; b < nb; ++b) { CK(hipFree(J[b])); CK(hipFree(Y[b])); }
    }
    return 0;
}
