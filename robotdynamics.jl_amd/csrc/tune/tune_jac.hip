// tune_jac.hip -- developer tool: within-process A/B timing of flagship Jacobian kernel variants
// (Satellite world-frame, RK4, ForwardAD, fp64, N = 65536).  Build: hipcc ... tune/tune_jac.hip -o tune_jac ; run on the GPU box.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "harness_tune.hpp"
#include "experimental_kernels.hpp"

using namespace rd;
#ifndef TUNE_DIAGJ
#define TUNE_DIAGJ true
#endif
using Mdl = RigidBody<double, WRENCH_SATELLITE, false, TUNE_DIAGJ>;
using Fn = Fun<Mdl, RULE_RK4>;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Variant { const char* name; void (*launch)(const Fn&, long long, const double*, const double*, const double*, double*, double*, hipStream_t); };

template <class Plan> void launch_plan(const Fn& fn, long long N, const double* Z, const double* t, const double* dt, double* J, double* Y, hipStream_t s) {
    constexpr size_t lds = JacLds<Fn, double, Plan>::BYTES;
    static bool init = false;
    if (!init) { CK(hipFuncSetAttribute((const void*)jac_ad_unrolled_kernel<Fn, double, Plan>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); init = true; }
    const long long ntiles = (N + TILE - 1) / TILE;
    int per_cu = (int)(160 * 1024 / lds); if (per_cu > 2048 / (Plan::NG * TILE)) per_cu = 2048 / (Plan::NG * TILE); if (per_cu < 1) per_cu = 1;
    const long long g = 256LL * per_cu;
    hipLaunchKernelGGL((jac_ad_unrolled_kernel<Fn, double, Plan>), dim3((unsigned)(ntiles < g ? ntiles : g)), dim3(Plan::NG * TILE), lds, s, fn, N, Z, t, dt, J, Y);
}

template <class Plan, int NC> void launch_split(const Fn& fn, long long N, const double* Z, const double* t, const double* dt, double* J, double* Y, hipStream_t s) {
    constexpr size_t lds = SplitLds<Fn, double, Plan, NC>::BYTES;
    static bool init = false;
    if (!init) { CK(hipFuncSetAttribute((const void*)jac_ad_split_kernel<Fn, double, Plan, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); init = true; }
    const long long ntiles = (N + TILE - 1) / TILE;
    hipLaunchKernelGGL((jac_ad_split_kernel<Fn, double, Plan, NC>), dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3((Plan::NG + NC) * TILE), lds, s, fn, N, Z, t, dt, J, Y, (unsigned*)nullptr);
}

template <class Plan, int NC> void launch_split2(const Fn& fn, long long N, const double* Z, const double* t, const double* dt, double* J, double* Y, hipStream_t s) {
    using L2 = Split2Lds<Fn, double, Plan>;
    constexpr size_t lds = L2::BYTES;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool init = false;
    if (!init) { CK(hipFuncSetAttribute((const void*)jac_ad_split2_kernel<Fn, double, Plan, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); init = true; }
    const long long ntiles = (N + TILE - 1) / TILE;
    hipLaunchKernelGGL((jac_ad_split2_kernel<Fn, double, Plan, NC>), dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3((Plan::NG + NC) * TILE), lds, s, fn, N, Z, t, dt, J, Y);
}
template <class Plan> void launch_split3(const Fn& fn, long long N, const double* Z, const double* t, const double* dt, double* J, double* Y, hipStream_t s) {
    using L3 = Split3Lds<Fn, double, Plan>;
    constexpr size_t lds = L3::BYTES;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool init = false;
    if (!init) { CK(hipFuncSetAttribute((const void*)jac_ad_split3_kernel<Fn, double, Plan>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); init = true; }
    const long long ntiles = (N + TILE - 1) / TILE;
    hipLaunchKernelGGL((jac_ad_split3_kernel<Fn, double, Plan>), dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3((Plan::NG + 4) * TILE), lds, s, fn, N, Z, t, dt, J, Y);
}
// two-chunk plans: chunk 0 = r,q,v,w10,w11 (cols 0-11), chunk 1 = w12,uF,tau (+ f(z)).  wave table per chunk:
//   chunk 0: q3,q4 -> 0 ; q5,q6 -> 1 ; w10 (+r) -> 2 ; w11 (+v) -> 3        chunk 1: w12 -> 0 ; tau16,uF13 -> 1 ; tau17,uF14 -> 2 ; tau18,uF15 (+y) -> 3
using Sat2A = TablePlan2<19, 4, 12, 3, /*r*/ 2, 2, 2, /*q*/ 0, 0, 1, 1, /*v*/ 3, 3, 3, /*w*/ 2, 3, 0, /*uF*/ 1, 2, 3, /*tau*/ 1, 2, 3>;
// same, but f(z) and all uF columns with the lightest chunk-1 wave pair
using Sat2C = TablePlan2<19, 4, 12, 3, /*r*/ 2, 2, 2, /*q*/ 0, 0, 1, 1, /*v*/ 3, 3, 3, /*w*/ 2, 3, 0, /*uF*/ 3, 3, 3, /*tau*/ 1, 2, 2>;
// chunk 0 = cols 0-9 (r,q,v), chunk 1 = w,uF,tau
using Sat2B = TablePlan2<19, 4, 10, 3, /*r*/ 0, 0, 0, /*q*/ 0, 1, 2, 3, /*v*/ 1, 2, 3, /*w*/ 0, 1, 2, /*uF*/ 3, 3, 3, /*tau*/ 3, 0, 1>;
template <class Plan, int SL, int OCC> void launch_wave(const Fn& fn, long long N, const double* Z, const double* t, const double* dt, double* J, double* Y, hipStream_t s) {
    const long long ntiles = (N + TILE - 1) / TILE;
    const long long grid = (ntiles + 7) / 8 * 8 * Plan::NGROUPS;
    using WL = WaveLds<Fn, double, Plan, SL>;
    constexpr unsigned lds = (unsigned)WL::BYTES;
    hipLaunchKernelGGL((jac_ad_wave_kernel<Fn, double, Plan, SL, OCC>), dim3((unsigned)grid), dim3(TILE), lds, s, fn, N, Z, t, dt, J, Y);
}
// hand-balanced satellite plans (columns: r 0-2, q 3-6, v 7-9, w 10-12, uF 13-15, tau 16-18)
using SatPlanA = TablePlan<19, 4, 2, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3>;
using SatPlanB = TablePlan<19, 4, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 2>;   // q | w10,w11 | w12,uF,tau18 | tau16,tau17 (+y)
using SatPlan5 = TablePlan<19, 5, 4, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 2, 2, 3, 4, 4, 4, 3, 4, 4>;   // 5 compute waves
// first-tile early part (TablePlanS): SPLIT = 10 (r, q, v)
using SatS0 = TablePlanS<19, 4, 2, 10, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3>;                 // the shipped table, early part = what waves 0 / 2 finish first
using SatS1 = TablePlanS<19, 4, 0, 10, /*r*/ 2, 2, 2, /*q*/ 0, 1, 2, 3, /*v*/ 2, 2, 2, /*w*/ 0, 1, 2, /*uF*/ 1, 0, 1, /*tau*/ 3, 3, 2>;
using SatP1 = TablePlan<19, 4, 0, /*r*/ 2, 2, 2, /*q*/ 0, 1, 2, 3, /*v*/ 2, 2, 2, /*w*/ 0, 1, 2, /*uF*/ 1, 0, 1, /*tau*/ 3, 3, 2>;
using SatS2 = TablePlanS<19, 4, 0, 10, /*r*/ 0, 0, 0, /*q*/ 0, 1, 2, 3, /*v*/ 1, 1, 1, /*w*/ 0, 1, 2, /*uF*/ 2, 2, 2, /*tau*/ 3, 3, 3>;
using SatP2 = TablePlan<19, 4, 0, /*r*/ 0, 0, 0, /*q*/ 0, 1, 2, 3, /*v*/ 1, 1, 1, /*w*/ 0, 1, 2, /*uF*/ 2, 2, 2, /*tau*/ 3, 3, 3>;
using SatS3 = TablePlanS<19, 4, 1, 10, /*r*/ 0, 0, 0, /*q*/ 0, 1, 2, 3, /*v*/ 1, 1, 1, /*w*/ 0, 1, 2, /*uF*/ 0, 1, 1, /*tau*/ 3, 3, 2>;
struct SatPlanBal8 {
    static constexpr int NGROUPS = 8, NRUNS = 2;
    static constexpr int B[8][2] = {{10, 0}, {11, 7}, {12, 0}, {14, 0}, {17, 0}, {18, 0}, {3, 0}, {5, 0}};
    static constexpr int E[8][2] = {{11, 3}, {12, 10}, {14, 0}, {17, 0}, {18, 0}, {19, 0}, {5, 0}, {7, 0}};
    static constexpr int run_begin(int g, int r) { return B[g][r]; }
    static constexpr int run_end(int g, int r) { return E[g][r]; }
    static constexpr int value_group = 4;
};
struct SatPlanCont8 {   // contiguous, cost-aware boundaries
    static constexpr int NGROUPS = 8, NRUNS = 1;
    static constexpr int Bd[9] = {0, 4, 6, 10, 11, 12, 14, 17, 19};
    static constexpr int run_begin(int g, int) { return Bd[g]; }
    static constexpr int run_end(int g, int) { return Bd[g + 1]; }
    static constexpr int value_group = 0;
};
struct SatPlanCont10 {
    static constexpr int NGROUPS = 10, NRUNS = 1;
    static constexpr int Bd[11] = {0, 4, 6, 10, 11, 12, 14, 16, 17, 18, 19};
    static constexpr int run_begin(int g, int) { return Bd[g]; }
    static constexpr int run_end(int g, int) { return Bd[g + 1]; }
    static constexpr int value_group = 0;
};
struct SatPlan4x2 {     // 4 groups, two sequential runs each (less primal replication)
    static constexpr int NGROUPS = 4, NRUNS = 2;
    static constexpr int B[4][2] = {{0, 3}, {5, 7}, {11, 13}, {15, 17}};
    static constexpr int E[4][2] = {{3, 5}, {7, 11}, {13, 15}, {17, 19}};
    static constexpr int run_begin(int g, int r) { return B[g][r]; }
    static constexpr int run_end(int g, int r) { return E[g][r]; }
    static constexpr int value_group = 0;
};

// pure streaming-store references: what the HBM write path sustains for the same 136 MB
__global__ void __launch_bounds__(256) fill_b64(double* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void __launch_bounds__(256) fill_b128(double2* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = make_double2((double)i, 1.0);
}
void launch_fill64(const Fn&, long long N, const double*, const double*, const double*, double* J, double*, hipStream_t s) {
    hipLaunchKernelGGL(fill_b64, dim3(2048), dim3(256), 0, s, J, N * 247);
}
void launch_fill128(const Fn&, long long N, const double*, const double*, const double*, double* J, double*, hipStream_t s) {
    hipLaunchKernelGGL(fill_b128, dim3(2048), dim3(256), 0, s, (double2*)J, N * 247 / 2);
}
void launch_fill128_big(const Fn&, long long N, const double*, const double*, const double*, double* J, double*, hipStream_t s) {
    hipLaunchKernelGGL(fill_b128, dim3((unsigned)(N * 247 / 2 / 256)), dim3(256), 0, s, (double2*)J, N * 247 / 2);
}

// experiment: stage the whole 64-knot J tile (64 x 1976 B, contiguous in global memory) in LDS, then copy it out
// linearly with 16-B stores (full 128-B lines per instruction).  Persistent grid, NW waves per block.
template <int NW>
__global__ void __launch_bounds__(NW * 64) tile_copy_kernel(long long ntiles, const double* __restrict__ Z, double* __restrict__ J) {
    extern __shared__ double tile[];   // 64*247 doubles
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const double zz = Z[tl * 64 * 19 + lane * 19];
        // every wave writes its share of columns for its lane's knot: element (knot=lane, c, i) at lane*247 + c*13 + i
        for (int c = w; c < 19; c += NW)
#pragma unroll
            for (int i = 0; i < 13; ++i) tile[lane * 247 + c * 13 + i] = zz + (double)(c * 13 + i);
        __syncthreads();
        double2* dst = (double2*)(J + tl * 64 * 247);
        const double2* src = (const double2*)tile;
        for (int u = threadIdx.x; u < 64 * 247 / 2; u += NW * 64) dst[u] = src[u];
        __syncthreads();
    }
}
// same, but the copy-out goes chunk by chunk over groups of CC columns (runs of CC*104 B per knot, 8-B stores):
// isolates the cost of partial-line writes at run boundaries.
template <int NW, int CC>
__global__ void __launch_bounds__(NW * 64) tile_copy_chunks_kernel(long long ntiles, const double* __restrict__ Z, double* __restrict__ J) {
    extern __shared__ double tile[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const double zz = Z[tl * 64 * 19 + lane * 19];
        for (int c = w; c < 19; c += NW)
#pragma unroll
            for (int i = 0; i < 13; ++i) tile[lane * 247 + c * 13 + i] = zz + (double)(c * 13 + i);
        __syncthreads();
        double* dst = J + tl * 64 * 247;
        for (int c0 = 0; c0 < 19; c0 += CC) {
            const int ncol = (19 - c0) < CC ? (19 - c0) : CC;
            const int run = ncol * 13;                       // doubles per knot in this chunk
            for (int u = threadIdx.x; u < 64 * run; u += NW * 64) {
                const int kn = u / run, o = u - kn * run;
                dst[kn * 247 + c0 * 13 + o] = tile[kn * 247 + c0 * 13 + o];
            }
        }
        __syncthreads();
    }
}
template <int NW, int CC> void launch_tilechunks(const Fn&, long long N, const double* Z, const double*, const double*, double* J, double*, hipStream_t s) {
    static bool init = false;
    if (!init) { CK(hipFuncSetAttribute((const void*)tile_copy_chunks_kernel<NW, CC>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 247 * 8)); init = true; }
    hipLaunchKernelGGL((tile_copy_chunks_kernel<NW, CC>), dim3(256), dim3(NW * 64), 64 * 247 * 8, s, N / 64, Z, J);
}
template <int NW> void launch_tilecopy(const Fn&, long long N, const double* Z, const double*, const double*, double* J, double*, hipStream_t s) {
    static bool init = false;
    if (!init) { CK(hipFuncSetAttribute((const void*)tile_copy_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 247 * 8)); init = true; }
    hipLaunchKernelGGL((tile_copy_kernel<NW>), dim3(256), dim3(NW * 64), 64 * 247 * 8, s, N / 64, Z, J);
}
template <int NW> void launch_tilecopy_np(const Fn&, long long N, const double* Z, const double*, const double*, double* J, double*, hipStream_t s) {
    static bool init = false;
    if (!init) { CK(hipFuncSetAttribute((const void*)tile_copy_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 247 * 8)); init = true; }
    hipLaunchKernelGGL((tile_copy_kernel<NW>), dim3((unsigned)(N / 64)), dim3(NW * 64), 64 * 247 * 8, s, N / 64, Z, J);
}

int main(int argc, char** argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 65536;
    const int NZ = 19, NX = 13, NBUF = 4, REPS = 30;
    std::vector<double> Z(N * NZ), t(N), dt(N, 0.1);
    unsigned long long st = 88172645463325252ULL;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (long long k = 0; k < N; ++k) {
        double q[4], nn = 0;
        for (int i = 0; i < 4; ++i) { double u1 = rnd() + 1e-12, u2 = rnd(); q[i] = std::sqrt(-2 * std::log(u1)) * std::cos(6.283185307179586 * u2); nn += q[i] * q[i]; }
        for (int i = 0; i < NZ; ++i) Z[k * NZ + i] = rnd();
        for (int i = 0; i < 4; ++i) Z[k * NZ + 3 + i] = q[i] / std::sqrt(nn);
        t[k] = 0.1 * k;
    }
    double p[10] = {1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    Fn fn{Mdl::make(p)};
    double *dZ, *dt_, *ddt, *dJ[NBUF], *dY[NBUF], *dJref;
    CK(hipMalloc(&dZ, N * NZ * 8)); CK(hipMalloc(&dt_, N * 8)); CK(hipMalloc(&ddt, N * 8)); CK(hipMalloc(&dJref, N * NX * NZ * 8));
    for (int b = 0; b < NBUF; ++b) { CK(hipMalloc(&dJ[b], N * NX * NZ * 8)); CK(hipMalloc(&dY[b], N * NX * 8)); }
    CK(hipMemcpy(dZ, Z.data(), N * NZ * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt_, t.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(ddt, dt.data(), N * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<Variant> vars = {
        TUNE_VARIANTS
    };
    std::vector<double> ref(N * NX * NZ), out(N * NX * NZ);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)N * (NZ * 8 + 16 + NX * NZ * 8 + NX * 8);
    for (int round = 0; round < 2; ++round)
    for (size_t v = 0; v < vars.size(); ++v) {
        for (int i = 0; i < 5; ++i) vars[v].launch(fn, N, dZ, dt_, ddt, dJ[i % NBUF], dY[i % NBUF], s);
        CK(hipStreamSynchronize(s));
        std::vector<float> ms(REPS);
        for (int i = 0; i < REPS; ++i) {
            CK(hipEventRecord(e0, s)); vars[v].launch(fn, N, dZ, dt_, ddt, dJ[i % NBUF], dY[i % NBUF], s); CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[i], e0, e1));
        }
        CK(hipGetLastError());
        std::sort(ms.begin(), ms.end());
        CK(hipMemcpy(out.data(), dJ[0], N * NX * NZ * 8, hipMemcpyDeviceToHost));
        if (v == 0 && round == 0) ref = out;
        double md = 0; for (size_t i = 0; i < out.size(); ++i) md = std::max(md, std::fabs(out[i] - ref[i]));
        printf("round %d  %-28s median %7.2f us  min %7.2f us   %6.0f GB/s (median)  frac %.3f   maxdiff_vs_v0 %.2e\n", round, vars[v].name, ms[REPS / 2] * 1e3, ms[0] * 1e3,
               bytes / (ms[REPS / 2] * 1e-3) / 1e9, bytes / (ms[REPS / 2] * 1e-3) / 8e12, md);
    }
    // back-to-back launches (what a sustained caller sees; the event-timed single launches above carry the launch latency)
    for (int round = 0; round < 2; ++round)
    for (size_t v = 0; v < vars.size(); ++v) {
        const int B2B = 400;
        for (int i = 0; i < 50; ++i) vars[v].launch(fn, N, dZ, dt_, ddt, dJ[i % NBUF], dY[i % NBUF], s);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < B2B; ++i) vars[v].launch(fn, N, dZ, dt_, ddt, dJ[i % NBUF], dY[i % NBUF], s);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float msb; CK(hipEventElapsedTime(&msb, e0, e1));
        printf("b2b   %d  %-28s %7.2f us per launch   frac %.3f\n", round, vars[v].name, msb * 1e3 / B2B, bytes / (msb * 1e-3 / B2B) / 8e12);
    }
#ifdef RD_STAMP
    {   // phase time stamps of variant 0 (a split-role kernel with 8 waves: compute wave 0 and copy wave 4 are printed)
#ifndef STAMP_EV
#define STAMP_EV 4      // events per tile
#endif
        long long* st; const int nslots = 256 * 8; CK(hipMalloc(&st, nslots * 64 * 8)); CK(hipMemset(st, 0, nslots * 64 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(rd::g_stamps), &st, sizeof st));
        for (int i = 0; i < 8; ++i) vars[0].launch(fn, N, dZ, dt_, ddt, dJ[i % NBUF], dY[i % NBUF], s);
        CK(hipStreamSynchronize(s));
        std::vector<long long> h(nslots * 64); CK(hipMemcpy(h.data(), st, nslots * 64 * 8, hipMemcpyDeviceToHost));
        long long t0 = -1; for (int b = 0; b < 256; ++b) { long long v = h[(b * 8 + 0) * 64 + 0]; if (v && (t0 < 0 || v < t0)) t0 = v; }
        auto us = [&](long long v) { return v ? (v - t0) / 100.0 : -1.0; };
        const int NE = STAMP_EV * 4;
        printf("phase stamps (us since first compute start), %d events per tile; C = compute wave 0..3, K = copy wave 4\n", STAMP_EV);
        for (int b : {0, 1, 37, 128, 255}) {
            for (int w = 0; w < 4; ++w) { printf(" blk %3d C%d:", b, w); for (int e = 0; e < NE; ++e) printf(" %6.2f", us(h[(b * 8 + w) * 64 + e])); printf("\n"); }
            printf("         K4:"); for (int e = 0; e < NE; ++e) printf(" %6.2f", us(h[(b * 8 + 4) * 64 + e])); printf("  done %6.2f\n", us(h[(b * 8 + 4) * 64 + 60]));
            printf("         K6:"); for (int e = 0; e < NE; ++e) printf(" %6.2f", us(h[(b * 8 + 6) * 64 + e])); printf("\n");
        }
        std::vector<double> done; std::vector<double> dsum(STAMP_EV, 0.0), ksum(STAMP_EV, 0.0); int cn = 0;
        for (int b = 0; b < 256; ++b) {
            for (int it = 0; it < 4; ++it) {
                const long long* c = &h[(b * 8 + 0) * 64 + it * STAMP_EV]; const long long* k = &h[(b * 8 + 4) * 64 + it * STAMP_EV];
                bool ok = true; for (int e = 0; e < STAMP_EV; ++e) ok = ok && c[e] && k[e];
                if (!ok) continue;
                for (int e = 1; e < STAMP_EV; ++e) { dsum[e] += (c[e] - c[e - 1]) / 100.0; ksum[e] += (k[e] - k[e - 1]) / 100.0; }
                ++cn;
            }
            done.push_back(us(h[(b * 8 + 4) * 64 + 60]));
        }
        std::sort(done.begin(), done.end());
        printf("mean event-to-event (us), compute wave 0:"); for (int e = 1; e < STAMP_EV; ++e) printf(" %.2f", dsum[e] / cn);
        printf(" | copy wave 4:"); for (int e = 1; e < STAMP_EV; ++e) printf(" %.2f", ksum[e] / cn); printf("  (n=%d)\n", cn);
        printf("block completion (all stores done): min %.2f p10 %.2f med %.2f p90 %.2f max %.2f\n", done[0], done[25], done[128], done[230], done[255]);
    }
#endif
    return 0;
}
