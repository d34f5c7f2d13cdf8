// tune_stamp.hip -- developer tool: where a split-role launch spends its time.  Built with -DRD_SPLIT_STAMP, the tile loop (kernels.hpp jac_split_loop) leaves
// 100 MHz wall-clock stamps per workgroup and iteration; this prints their averages over the workgroups for ONE launch in a stream of identical ones.
//   -DKIND=0 satellite RK4 ForwardAD (flagship)   1 satellite ImplicitMidpoint + IFT   2 body-frame satellite RK4   3 body-frame quadrotor RK4
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
#ifndef KIND
#define KIND 0
#endif
#if KIND == 3
using Mdl = RigidBody<double, WRENCH_QUADROTOR, true, true, true>;
constexpr int NU_ = 4;
#elif KIND == 2
using Mdl = RigidBody<double, WRENCH_SATELLITE, true, true>;
constexpr int NU_ = 6;
#else
using Mdl = RigidBody<double, WRENCH_SATELLITE, false, true>;
constexpr int NU_ = 6;
#endif
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
    constexpr int NZ = 13 + NU_, OW = 13 * NZ;
    std::vector<double> hz((size_t)N * NZ), ht(N), hdt(N, 0.1);
    for (long long k = 0; k < N; ++k) {
        for (int i = 0; i < NZ; ++i) hz[k * NZ + i] = 0.1 + 0.8 * ((k * NZ + i) * 2654435761u % 1000) / 1000.0;
        double q2 = 0; for (int i = 3; i < 7; ++i) q2 += hz[k * NZ + i] * hz[k * NZ + i];
        for (int i = 3; i < 7; ++i) hz[k * NZ + i] /= std::sqrt(q2);
        ht[k] = 0.1 * k;
    }
    double *Z, *t, *dt, *J[4], *Y[4]; int* flag; long long* st;
    const int NWG = 256;
    CK(hipMalloc(&Z, N * NZ * 8)); CK(hipMalloc(&t, N * 8)); CK(hipMalloc(&dt, N * 8)); CK(hipMalloc(&flag, (2 + N / 64) * 4)); CK(hipMemset(flag, 0, (2 + N / 64) * 4));
    CK(hipMalloc(&st, (size_t)NWG * 8 * 16 * 8)); CK(hipMemset(st, 0, (size_t)NWG * 8 * 16 * 8));
    for (int b = 0; b < 4; ++b) { CK(hipMalloc(&J[b], N * OW * 8)); CK(hipMalloc(&Y[b], N * 13 * 8)); }
    CK(hipMemcpy(Z, hz.data(), hz.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), hdt.size() * 8, hipMemcpyHostToDevice));
#if KIND == 3
    double prm[16] = {0.5, 0.0023, 0, 0, 0, 0.0023, 0, 0, 0, 0.004, 0, 0, -9.81, 0.1750, 1.0, 0.0245};
#else
    double prm[10] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
#endif
#if KIND == 1
    Fn fn{Mdl::make(prm), 1e-12, flag, 10, 2, nullptr};
    auto k = implicit_ift_split_kernel<Fn, double, Plan, 4>;
#else
    Fn fn{Mdl::make(prm)};
    auto k = jac_ad_split_kernel<Fn, double, Plan, 4>;
#endif
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    constexpr size_t lds = SplitLds<Fn, double, Plan, 4>::BYTES;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int i) { hipLaunchKernelGGL(k, dim3(NWG), dim3(512), lds, s, fn, N, Z, t, dt, J[i % 4], Y[i % 4], (unsigned*)nullptr); };
    for (int i = 0; i < 2000; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 400; ++i) launch(i);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("KIND %d  N %lld  unstamped-pointer launches: %.2f us each\n", KIND, N, ms * 1e3 / 400);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_split_stamp), &st, sizeof st));
    for (int i = 0; i < 50; ++i) launch(i);           // the stamped launch sits in the middle of a stream of identical ones
    CK(hipStreamSynchronize(s));
    std::vector<long long> h((size_t)NWG * 8 * 16);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    auto S = [&](int wg, int it, int sl) { return h[((size_t)wg * 8 + it) * 16 + sl]; };
    long long t0 = 1LL << 62, t1 = 0;
    for (int wg = 0; wg < NWG; ++wg) for (int it = 0; it < 8; ++it) for (int sl = 0; sl < 16; ++sl) { long long v = S(wg, it, sl); if (v) { if (v < t0) t0 = v; if (v > t1) t1 = v; } }
    printf("first stamp -> last stamp of the launch: %.2f us (stamps are 10 ns)\n", (t1 - t0) * 0.01);
    { double pro = 0, epi = 0, epc = 0; int n = 0; const int last = (int)((N / 64 + NWG - 1) / NWG) - 1;
      for (int wg = 0; wg < NWG; ++wg) if (S(wg, 0, 13) && S(wg, 0, 15) && last < 8) { ++n; pro += S(wg, 0, 0) - S(wg, 0, 13); epi += S(wg, 0, 15) - S(wg, last, 3); epc += S(wg, 0, 15) - S(wg, last, 10); }
      if (n) printf("workgroup entry -> first iteration %.2f us;  last barrier B -> compute wave 0 leaves %.2f us (copy wave 0's last store issued -> that: %.2f us)\n", pro / n * 0.01, epi / n * 0.01, epc / n * 0.01); }
    {   // spread over the workgroups: when does a workgroup enter / issue its last store, by XCD (workgroup index % 8)
        const int last = (int)((N / 64 + NWG - 1) / NWG) - 1;
        if (last < 8) {
            std::vector<double> en, ex;
            double xin[8] = {0}, xout[8] = {0}; int xn[8] = {0};
            for (int wg = 0; wg < NWG; ++wg) if (S(wg, 0, 13) && S(wg, last, 10)) {
                const double a = (S(wg, 0, 13) - t0) * 0.01, b = (S(wg, last, 10) - t0) * 0.01;
                en.push_back(a); ex.push_back(b); xin[wg & 7] += a; xout[wg & 7] += b; ++xn[wg & 7];
            }
            std::sort(en.begin(), en.end()); std::sort(ex.begin(), ex.end());
            auto pc = [](const std::vector<double>& v, double q) { return v.empty() ? 0.0 : v[(size_t)(q * (v.size() - 1))]; };
            printf("workgroup entry (us after the first stamp): min %.2f  median %.2f  90%% %.2f  max %.2f\n", pc(en, 0), pc(en, 0.5), pc(en, 0.9), pc(en, 1));
            printf("last store issued:                          min %.2f  median %.2f  90%% %.2f  max %.2f\n", pc(ex, 0), pc(ex, 0.5), pc(ex, 0.9), pc(ex, 1));
            printf("by XCD (entry / last store issued):");
            for (int x = 0; x < 8; ++x) if (xn[x]) printf("  %.1f/%.1f", xin[x] / xn[x], xout[x] / xn[x]);
            printf("\n");
        }
    }
    const int ntw = (int)((N / 64 + NWG - 1) / NWG);
    printf("it |  start  | cmp w0   w1   w2   w3 | waitA | A->B |  lds  | issue | helper | iter\n");
    for (int it = 0; it < ntw && it < 8; ++it) {
        double a[12] = {0}; int n = 0;
        for (int wg = 0; wg < NWG; ++wg) {
            if (!S(wg, it, 0) || !S(wg, it, 3)) continue;
            ++n;
            a[0] += (S(wg, it, 0) - t0);
            for (int g = 0; g < 4; ++g) a[1 + g] += S(wg, it, 4 + g) - S(wg, it, 0);
            long long mx = 0; for (int g = 0; g < 4; ++g) mx = S(wg, it, 4 + g) > mx ? S(wg, it, 4 + g) : mx;
            a[5] += S(wg, it, 2) - mx;                   // last compute wave done -> barrier A passed (waiting for copy waves / helper)
            a[6] += S(wg, it, 3) - S(wg, it, 2);         // A -> B seen by compute wave 0
            a[7] += S(wg, it, 9) - S(wg, it, 8);         // copy wave 0: LDS readout
            a[8] += S(wg, it, 10) - S(wg, it, 9);        // copy wave 0: barrier B + store issue
            a[9] += S(wg, it, 12) ? (S(wg, it, 12) - S(wg, it, 11)) : 0;
            a[10] += S(wg, it, 3) - S(wg, it, 0);
        }
        if (!n) continue;
        printf("%2d | %7.2f | %5.2f %5.2f %5.2f %5.2f | %5.2f | %5.2f | %5.2f | %5.2f | %6.2f | %5.2f   (%d workgroups)\n", it, a[0] / n * 0.01, a[1] / n * 0.01, a[2] / n * 0.01, a[3] / n * 0.01,
               a[4] / n * 0.01, a[5] / n * 0.01, a[6] / n * 0.01, a[7] / n * 0.01, a[8] / n * 0.01, a[9] / n * 0.01, a[10] / n * 0.01, n);
    }
    return 0;
}
