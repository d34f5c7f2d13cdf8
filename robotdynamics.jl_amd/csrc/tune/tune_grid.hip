// tune_grid.hip -- developer tool: the flagship fp64 Jacobian kernel (satellite RK4 ForwardAD, J + x_next) on persistent grids of fewer workgroups than CUs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize tune/tune_grid.hip -o tune/bin/tune_grid
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
using TT = double;
using Mdl = RigidBody<TT, WRENCH_SATELLITE, false, true>;
using Fn = Fun<Mdl, RULE_RK4>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int NZ = 19, NX = 13, NSET = 4;
    double prm[16] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    Fn fn{Mdl::make(prm)};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto kj = jac_ad_split_kernel<Fn, TT, SatPlan, 4>;
    auto kd = jac_ad_split_kernel<Fn, TT, SatPlan, 4, true>;
    unsigned* sched; CK(hipMalloc(&sched, 64 * 64)); CK(hipMemset(sched, 0, 64 * 64));
    constexpr size_t ldsj = SplitLds<Fn, TT, SatPlan, 4>::BYTES;
    CK(hipFuncSetAttribute((const void*)kj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsj));
    CK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsj));
    for (long long N : {65536LL, 131072LL, 262144LL, 1048576LL}) {
        std::vector<TT> hz((size_t)N * NZ);
        for (long long k = 0; k < N; ++k) {
            for (int i = 0; i < NZ; ++i) hz[k * NZ + i] = (TT)(0.1 + 0.8 * ((k * NZ + i) * 2654435761u % 1000) / 1000.0);
            double nq = 0; for (int i = 3; i < 7; ++i) nq += hz[k * NZ + i] * hz[k * NZ + i];
            for (int i = 3; i < 7; ++i) hz[k * NZ + i] /= sqrt(nq);
        }
        std::vector<double> ht(N), hdt(N, 0.1);
        for (long long k = 0; k < N; ++k) ht[k] = 0.1 * k;
        TT *Z[NSET], *J[NSET], *Y[NSET]; double *t, *dt;
        const int nset = N > 300000 ? 2 : NSET;
        for (int i = 0; i < nset; ++i) { CK(hipMalloc(&Z[i], N * NZ * sizeof(TT))); CK(hipMalloc(&J[i], N * NX * NZ * sizeof(TT))); CK(hipMalloc(&Y[i], N * NX * sizeof(TT)));
                                         CK(hipMemcpy(Z[i], hz.data(), hz.size() * sizeof(TT), hipMemcpyHostToDevice)); }
        CK(hipMalloc(&t, N * 8)); CK(hipMalloc(&dt, N * 8));
        CK(hipMemcpy(t, ht.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), N * 8, hipMemcpyHostToDevice));
        const double alg = (double)N * 2248.0;
        const int reps = N > 300000 ? 30 : 300;
        {   // clocks and TLBs settle over ~100 ms of load
            int r = 0; const int nw = (int)(100000.0 / (N * 4.2e-4)) + 1;
            for (int i = 0; i < nw; ++i) { hipLaunchKernelGGL(kj, dim3(256), dim3(512), ldsj, s, fn, N, Z[r % nset], t, dt, J[r % nset], Y[r % nset], (unsigned*)nullptr); ++r; }
            CK(hipStreamSynchronize(s));
        }
        for (int pass = 0; pass < 2; ++pass)
        for (int dyn = 0; dyn < 2; ++dyn)
        for (int grid : {256, 224, 208, 192, 176, 160, 144, 256}) {
            const long long ntiles = (N + 63) / 64;
            if (dyn && ntiles < 8LL * grid) continue;
            int r = 0;
            auto launch = [&]() { if (dyn) hipLaunchKernelGGL(kd, dim3(grid), dim3(512), ldsj, s, fn, N, Z[r % nset], t, dt, J[r % nset], Y[r % nset], sched + (r % 32) * 16);
                                  else hipLaunchKernelGGL(kj, dim3(grid), dim3(512), ldsj, s, fn, N, Z[r % nset], t, dt, J[r % nset], Y[r % nset], (unsigned*)nullptr); ++r; };
            for (int i = 0; i < reps / 4; ++i) launch();
            CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("N %8lld %s grid %3d: %8.2f us per launch = %.3f of 8 TB/s\n", N, dyn ? "dyn   " : "static", grid, ms * 1e3 / reps, alg / (ms * 1e-3 / reps) / 8e12);
        }
        for (int i = 0; i < nset; ++i) { CK(hipFree(Z[i])); CK(hipFree(J[i])); CK(hipFree(Y[i])); }
        CK(hipFree(t)); CK(hipFree(dt));
    }
    return 0;
}
