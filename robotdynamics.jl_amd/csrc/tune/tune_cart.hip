// tune_cart.hip -- developer tool: time per step of the plain rollout kernel (one lane per trajectory) for the cartpole, fp32 / fp64
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "harness_tune.hpp"
#include "../dispatch.hpp"
using namespace rd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
template <class T> void run(const char* name) {
    using Fn = Fun<Cartpole<T>, RULE_RK4>;
    const long long M = 512, K = 256;
    std::vector<T> hz(M * K * 5);
    for (size_t i = 0; i < hz.size(); ++i) hz[i] = (T)(0.1 + 0.8 * ((i * 2654435761u) % 1000) / 1000.0);
    std::vector<double> ht(M * K, 0.0), hdt(M * K, 0.01);
    T* Z; double *t, *dt;
    CK(hipMalloc(&Z, hz.size() * sizeof(T))); CK(hipMalloc(&t, M * K * 8)); CK(hipMalloc(&dt, M * K * 8));
    CK(hipMemcpy(Z, hz.data(), hz.size() * sizeof(T), hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), M * K * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), M * K * 8, hipMemcpyHostToDevice));
    double prm[4] = {1.0, 0.2, 0.5, 9.81};
    Fn fn{Cartpole<T>::make(prm)};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() { hipLaunchKernelGGL((rollout_kernel<Fn, T>), dim3((unsigned)(M / 64)), dim3(64), 0, 0, fn, K, M, K, 1LL, Z, t, dt, (const T*)nullptr); };
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<T> out(5);
    CK(hipMemcpy(out.data(), Z + (K - 1) * 5, 5 * sizeof(T), hipMemcpyDeviceToHost));
    printf("%-10s sincosf variant %d: %8.2f us per rollout, %6.3f us per step   (x_K[0..1] = %.6f %.6f)\n", name, RD_SINCOSF, ms * 1e3 / 20, ms * 1e3 / 20 / (K - 1), (double)out[0], (double)out[1]);
}
int main() { run<float>("fp32"); run<double>("fp64"); return 0; }
