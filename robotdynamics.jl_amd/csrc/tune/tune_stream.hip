// tune_stream.hip -- developer tool: rollout_stream_kernel against rollout_kernel / rollout_cascade_kernel (results and time), both layouts.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize tune/tune_stream.hip -o tune/tune_stream
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "harness_tune.hpp"
#include "../dispatch.hpp"
#include "../rollout_stream.hpp"
using namespace rd;
#ifndef TUNE_T
#define TUNE_T double
#endif
using T = TUNE_T;
#ifdef TUNE_CART
using Mdl = Cartpole<T>;
constexpr int NZc = 5;
#else
using Mdl = RigidBody<T, WRENCH_SATELLITE, false, true>;
constexpr int NZc = 19;
#endif
using Fn = Fun<Mdl, RULE_RK4>;
typedef unsigned int V4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
template <bool TM, class U> void launch_stream(const Fn& fn, long long K, long long M, T* Z, const double* t, const double* dt, const T* X0) {
    using L = StreamCfg<Fn, T, TM>;
    auto kern = rollout_stream_kernel<Fn, T, TM, U>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES)); once = true; }
    kern<<<(unsigned)((M + 63) / 64), STREAM_THREADS, L::BYTES, 0>>>(fn, K, M, Z, t, dt, X0);
}
int main(int argc, char** argv) {
    double p[64] = {0}; p[0] = 1.0; p[1] = 1.0; p[5] = 2.0; p[9] = 3.0;
#ifdef TUNE_CART
    p[0] = 1.0; p[1] = 0.2; p[2] = 0.5; p[3] = 9.81;
#endif
    Fn fn{Mdl::make(p)};
    const long long shapes[][2] = {{100, 7}, {16384, 64}, {65536, 32}, {262144, 16}, {32768, 101}, {65537, 18}};
    for (auto& sh : shapes) {
        const long long M = sh[0], K = sh[1], NZ = NZc;
        std::vector<T> hz(M * K * NZ);
        for (size_t i = 0; i < hz.size(); ++i) hz[i] = (T)((double)((i * 2654435761u) % 1000) / 1000.0);
        std::vector<double> ht(M * K, 0.0), hdt(M * K, 0.01);
        for (long long j = 0; j < M; ++j) hdt[j * K + K - 1] = 0.0;
        T *Z, *Z2; double *t, *dt;
        const size_t zb = hz.size() * sizeof(T);
        CK(hipMalloc(&Z, zb)); CK(hipMalloc(&Z2, zb)); CK(hipMalloc(&t, ht.size() * 8)); CK(hipMalloc(&dt, ht.size() * 8));
        CK(hipMemcpy(t, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), ht.size() * 8, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<T> r0(hz.size()), r1(hz.size());
        for (int tm = 0; tm < 2; ++tm) {
            const long long sj = tm ? 1 : K, sk = tm ? M : 1;
            const bool al = (((tm ? M : K) * NZ * sizeof(T)) % 16) == 0;
            for (int mode = 0; mode < 2; ++mode) {
                auto launch = [&](T* z) {
                    if (mode == 0) rollout_kernel<Fn, T><<<(unsigned)((M + 63) / 64), 64, 0, 0>>>(fn, K, M, sj, sk, z, t, dt, (const T*)nullptr);
                    else if (tm) { if (al) launch_stream<true, V4>(fn, K, M, z, t, dt, nullptr); else launch_stream<true, T>(fn, K, M, z, t, dt, nullptr); }
                    else { if (al) launch_stream<false, V4>(fn, K, M, z, t, dt, nullptr); else launch_stream<false, T>(fn, K, M, z, t, dt, nullptr); }
                };
                CK(hipMemcpy(Z, hz.data(), zb, hipMemcpyHostToDevice));
                launch(Z); CK(hipDeviceSynchronize());
                CK(hipMemcpy((mode ? r1 : r0).data(), Z, zb, hipMemcpyDeviceToHost));
                for (int i = 0; i < 2; ++i) { launch(Z); launch(Z2); }
                CK(hipDeviceSynchronize());
                const int reps = 6;
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) launch(i & 1 ? Z2 : Z);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / reps, bytes = (double)M * K * (NZ * sizeof(T) + 16 + (NZ - (NZc == 19 ? 6 : 1)) * sizeof(T));
                printf("%-7s %-3s %s M=%-7lld K=%-4lld %9.2f us  frac %.3f\n", mode ? "stream" : "plain", tm ? "tm" : "tj", al ? "16B" : "el ", M, K, us, bytes / (us * 1e-6) / 8e12);
            }
            double md = 0; size_t bad = 0;
            for (size_t i = 0; i < r0.size(); ++i) { const double d = std::fabs((double)r0[i] - (double)r1[i]); if (!(d <= md)) { md = d; bad = i; } }
            printf("        max |stream - plain| = %.3e (at %zu)\n", md, bad);
        }
        CK(hipFree(Z)); CK(hipFree(Z2)); CK(hipFree(t)); CK(hipFree(dt));
    }
    return 0;
}
