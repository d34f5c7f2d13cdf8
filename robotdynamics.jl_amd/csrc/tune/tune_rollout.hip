// tune_rollout.hip -- developer tool: timing of the rollout kernels (cascade / plain; knot-contiguous and time-major).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize [-DRD_ROLL_ABLATE=n] tune/tune_rollout.hip -o tune/tune_rollout_n
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../dispatch.hpp"
using namespace rd;
#ifndef TUNE_T
#define TUNE_T double
#endif
using T = TUNE_T;
using Mdl = RigidBody<T, WRENCH_SATELLITE, false, true>;
using Fn = Fun<Mdl, RULE_RK4>;
using Cas = Cascade<Mdl>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const char* tag = argc > 1 ? argv[1] : "";
    double p[64] = {0}; p[0] = 1.0; p[1] = 1.0; p[5] = 2.0; p[9] = 3.0;
    Fn fn{Mdl::make(p)};
    const int shapes[][2] = {{64, 256}, {512, 256}, {4096, 101}, {16384, 64}};
    for (auto& sh : shapes) {
        const long long M = sh[0], K = sh[1], NZ = 19;
        std::vector<T> hz(M * K * NZ);
        for (size_t i = 0; i < hz.size(); ++i) hz[i] = (T)((double)((i * 2654435761u) % 1000) / 1000.0);
        std::vector<double> ht(M * K, 0.0), hdt(M * K, 0.01);
        T* Z; double *t, *dt;
        CK(hipMalloc(&Z, hz.size() * sizeof(T))); CK(hipMalloc(&t, ht.size() * 8)); CK(hipMalloc(&dt, ht.size() * 8));
        CK(hipMemcpy(Z, hz.data(), hz.size() * sizeof(T), hipMemcpyHostToDevice));
        CK(hipMemcpy(t, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), ht.size() * 8, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 4; ++mode) {
            constexpr int R = 4;
            auto launch = [&]() {
                constexpr size_t lds = CascadeLds<Fn, Cas, R>::BYTES;
                const unsigned grid = (unsigned)((M + 63) / 64);
                const long long sj = mode < 2 ? K : 1, sk = mode < 2 ? 1 : M;      // modes 2, 3: time-major
                if (mode % 2 == 0) rollout_cascade_kernel<Fn, T, Cas, R><<<grid, (Cas::NB + 1) * 64, lds, 0>>>(fn, K, M, sj, sk, Z, t, dt, (const T*)nullptr);
                else rollout_kernel<Fn, T><<<grid, 64, 0, 0>>>(fn, K, M, sj, sk, Z, t, dt, (const T*)nullptr);
            };
            CK(hipFuncSetAttribute((const void*)rollout_cascade_kernel<Fn, T, Cas, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CascadeLds<Fn, Cas, 4>::BYTES));
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            const int reps = 10;
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-8s %-10s M=%-6lld K=%-4lld %8.2f us/launch  %.4f us/step\n", tag, mode == 0 ? "cascade" : mode == 1 ? "plain" : mode == 2 ? "cascade-tm" : "plain-tm", M, K, ms * 1e3 / reps, ms * 1e3 / reps / (K - 1));
        }
        CK(hipFree(Z)); CK(hipFree(t)); CK(hipFree(dt));
    }
    return 0;
}
