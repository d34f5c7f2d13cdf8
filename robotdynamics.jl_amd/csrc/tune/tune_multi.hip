// tune_multi.hip -- developer tool: the flagship Jacobian kernel on one batch against the same tile loop on a batch table
// (rdhip_jacobian_multi), back-to-back launches, outputs rotated.  Build: hipcc ... -I.. tune/tune_multi.hip -o tune/tune_multi
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "harness_tune.hpp"
#include "../fam_rigid.hpp"
using namespace rd;
using Mdl = RigidBody<double, WRENCH_SATELLITE, false, true>;
using Fn = Fun<Mdl, RULE_RK4>;
using Plan = SatPlan;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    const long long NT = 65536 + 80000;                  // knot points in the pool (1024 x 101 = 103 424)
    const int NBUF = 4;
    std::vector<double> hz((size_t)NT * 19), ht(NT), hdt(NT, 0.1);
    for (long long k = 0; k < NT; ++k) {
        for (int i = 0; i < 19; ++i) hz[k * 19 + i] = 0.1 + 0.8 * ((k * 19 + i) * 2654435761u % 1000) / 1000.0;
        ht[k] = 0.1 * k;
    }
    double *Z, *t, *dt, *J[NBUF], *Y[NBUF];
    CK(hipMalloc(&Z, NT * 19 * 8)); CK(hipMalloc(&t, NT * 8)); CK(hipMalloc(&dt, NT * 8));
    for (int b = 0; b < NBUF; ++b) { CK(hipMalloc(&J[b], NT * 247 * 8)); CK(hipMalloc(&Y[b], NT * 13 * 8)); }
    CK(hipMemcpy(Z, hz.data(), NT * 19 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(t, ht.data(), NT * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, hdt.data(), NT * 8, hipMemcpyHostToDevice));
    double prm[10] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    Fn fn{Mdl::make(prm)};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr size_t lds = SplitLds<Fn, double, Plan, 4>::BYTES;
    auto ks = jac_ad_split_kernel<Fn, double, Plan, 4>;
    auto km = jac_split_multi_kernel<Fn, double, Plan, 4, AdUnrolledBody<Fn, double, Plan>, 1>;
    CK(hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)km, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto timeit = [&](const char* name, long long knots, auto&& launch) {
        for (int i = 0; i < 50; ++i) launch(i);
        CK(hipStreamSynchronize(s));
        double best = 1e30, sum = 0;
        for (int round = 0; round < 5; ++round) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps; sum += us; if (us < best) best = us;
        }
        printf("%-52s %8.2f us (best %8.2f)  frac %.3f\n", name, sum / 5, best, knots * 2248.0 / (sum / 5 * 1e-6) / 8e12);
    };
    timeit("single 65536", 65536, [&](int i) { hipLaunchKernelGGL(ks, dim3(256), dim3(512), lds, s, fn, 65536LL, Z, t, dt, J[i % NBUF], Y[i % NBUF], (unsigned*)nullptr); });
    timeit("single 103424", 103424, [&](int i) { hipLaunchKernelGGL(ks, dim3(256), dim3(512), lds, s, fn, 103424LL, Z, t, dt, J[i % NBUF], Y[i % NBUF], (unsigned*)nullptr); });
    struct Case { const char* name; int nb; long long per; bool table; };
    const Case cases[] = {{"multi inline 16 x 4096", 16, 4096, false}, {"multi inline 32 x 2048", 32, 2048, false}, {"multi table uniform 64 x 1024", 64, 1024, true},
                          {"multi table uniform 1024 x 101", 1024, 101, true}, {"multi table ragged 1024 x 101(+-)", 1024, -101, true}, {"multi inline 32 x 101", 32, 101, false}};
    for (const Case& c : cases) {
        MultiBatch<double> mb[NBUF];
        long long knots = 0;
        for (int q = 0; q < NBUF; ++q) {
            MultiBatch<double>& m = mb[q];
            std::memset(&m, 0, sizeof m);
            std::vector<MultiRec<double>> recs(c.nb); std::vector<unsigned> t0(c.nb + 1);
            long long acc = 0, off = 0; knots = 0;
            for (int b = 0; b < c.nb; ++b) {
                long long n = c.per > 0 ? c.per : (b % 3 == 0 ? 101 : b % 3 == 1 ? 64 : 138);
                recs[b] = MultiRec<double>{Z + off * 19, t + off, dt + off, J[q] + off * 247, Y[q] + off * 13, n, {0, 0}};
                t0[b] = (unsigned)acc; acc += (n + 63) / 64; off += (n + 1) / 2 * 2; knots += n;      // (every trajectory's arrays start 16-byte aligned, as the API requires)
            }
            t0[c.nb] = (unsigned)acc;
            m.nb = c.nb; m.total = acc; m.group = 1;
            if (c.table) {
                void* d; CK(hipMalloc(&d, recs.size() * 64 + t0.size() * 4));
                CK(hipMemcpy(d, recs.data(), recs.size() * 64, hipMemcpyHostToDevice));
                CK(hipMemcpy((char*)d + recs.size() * 64, t0.data(), t0.size() * 4, hipMemcpyHostToDevice));
                m.ext = (const MultiRec<double>*)d; m.ext_tile0 = (const unsigned*)((char*)d + recs.size() * 64);
                m.uniform = c.per > 0 ? (int)((c.per + 63) / 64) : 0;
                m.group = (c.nb + 31) / 32;
                for (int i = 0; i <= 32; ++i) m.tile0[i] = (long long)i * m.group < c.nb ? t0[(size_t)i * m.group] : (unsigned)acc;
            } else {
                for (int i = 0; i <= 32; ++i) m.tile0[i] = i < c.nb ? t0[i] : (unsigned)acc;
                for (int i = 0; i < c.nb; ++i) { m.N[i] = recs[i].N; m.Z[i] = recs[i].Z; m.t[i] = recs[i].t; m.dt[i] = recs[i].dt; m.J[i] = recs[i].J; m.Y[i] = recs[i].Y; }
            }
        }
        const unsigned grid = (unsigned)(mb[0].total < 256 ? mb[0].total : 256);
        timeit(c.name, knots, [&](int i) { hipLaunchKernelGGL(km, dim3(grid), dim3(512), lds, s, fn, mb[i % NBUF], (unsigned*)nullptr); });
        if (c.nb == 1024) { char nm[100]; snprintf(nm, sizeof nm, "%s grid 255", c.name); timeit(nm, knots, [&](int i) { hipLaunchKernelGGL(km, dim3(255), dim3(512), lds, s, fn, mb[i % NBUF], (unsigned*)nullptr); }); }
    }
    // correctness spot check: single vs multi(16 x 4096) on buffer 0 / 1
    CK(hipStreamSynchronize(s));
    printf("done\n");
    return 0;
}
