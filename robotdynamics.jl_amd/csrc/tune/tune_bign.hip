// tune_bign.hip -- developer tool: the fp64 flagship kernel at 256 K ... 4 M knot points under different tile -> workgroup orders (the 2 M - 4 M dip of
// DESIGN 4.5.1): plain, and R = 1 / 2 / 4 contiguous ranges per XCD, adjacent or interleaved.  Outputs rotate over two buffers.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize tune/tune_bign.hip -o tune/tune_bign
#include <cstdio>
#include <cstdlib>
#include <vector>
#define RD_TILE_MAP_RUNTIME 1
// (rd::g_tile_map: harness_tune.hpp)
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
using TT = double;
using Mdl = RigidBody<TT, WRENCH_SATELLITE, false, true>;
using Fn = Fun<Mdl, RULE_RK4>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const long long NZ = 19, NX = 13;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto kj = jac_ad_split_kernel<Fn, TT, SatPlan, 4>;
    constexpr size_t ldsj = SplitLds<Fn, TT, SatPlan, 4>::BYTES;
    CK(hipFuncSetAttribute((const void*)kj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsj));
    double prm[16] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    Fn fn{Mdl::make(prm)};
    const long long NMAX = 4LL << 20;
    std::vector<TT> hz((size_t)NMAX * NZ);
    for (long long k = 0; k < NMAX; ++k) for (int i = 0; i < NZ; ++i) hz[k * NZ + i] = (TT)(0.1 + 0.8 * ((k * NZ + i) * 2654435761u % 1000) / 1000.0);
    std::vector<double> ht(NMAX, 0.0), hdt(NMAX, 0.01);
    TT *Z, *Z2, *J[2], *Y[2]; double *t, *dt, *t2, *dt2;
    CK(hipMalloc(&Z, NMAX * NZ * sizeof(TT))); CK(hipMalloc(&t, NMAX * 8)); CK(hipMalloc(&dt, NMAX * 8));
    CK(hipMalloc(&Z2, NMAX * NZ * sizeof(TT))); CK(hipMalloc(&t2, NMAX * 8)); CK(hipMalloc(&dt2, NMAX * 8));
    for (int b = 0; b < 2; ++b) { CK(hipMalloc(&J[b], NMAX * NX * NZ * sizeof(TT))); CK(hipMalloc(&Y[b], NMAX * NX * sizeof(TT))); }
    CK(hipMemcpy(Z, hz.data(), hz.size() * sizeof(TT), hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), NMAX * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), NMAX * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(Z2, Z, NMAX * NZ * sizeof(TT), hipMemcpyDeviceToDevice)); CK(hipMemcpy(t2, t, NMAX * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(dt2, dt, NMAX * 8, hipMemcpyDeviceToDevice));
    const bool rot_in = argc > 2 && atoi(argv[2]) != 0;       // second argument: rotate the INPUT arrays over two copies as well
    printf("inputs %s\n", rot_in ? "rotate over two copies" : "one copy (re-read by every launch)");
    const int maps[] = {0, 100};
    const int rounds = argc > 1 ? atoi(argv[1]) : 2;
    for (int rnd = 0; rnd < rounds; ++rnd)
    for (long long N : {262144LL, 524288LL, 1048576LL, 2097152LL, 4194304LL}) {
        const double alg = (double)N * (NZ * sizeof(TT) + 16 + NX * NZ * sizeof(TT) + NX * sizeof(TT));
        printf("N=%-8lld", N);
        for (int mp : maps) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(rd::g_tile_map), &mp, sizeof(int)));
            const int reps = (int)((8LL << 20) / N) + 4;
            auto launch = [&](int i) { const bool o = rot_in && (i & 1); hipLaunchKernelGGL(kj, dim3(256), dim3(512), ldsj, s, fn, N, o ? Z2 : Z, o ? t2 : t, o ? dt2 : dt, J[i & 1], Y[i & 1], (unsigned*)nullptr); };
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  map %3d: %8.1f us %.3f", mp, ms * 1e3 / reps, alg / (ms * 1e-3 / reps) / 8e12);
        }
        printf("\n");
    }
    return 0;
}
