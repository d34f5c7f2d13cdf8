// fp64_ubench.hip -- developer tool: fp64 VALU issue rate / dependent latency on gfx950, and streaming-store ceilings.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_ubench.hip -o tools/fp64_ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int ILP, int OP>
__global__ void __launch_bounds__(256) fma_chain(double* out, int iters, double a, double b) {
    double acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (OP == 0) acc[i] = __builtin_fma(acc[i], a, b);
                else if (OP == 1) acc[i] = acc[i] * a;
                else acc[i] = acc[i] + b;
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
}

template <int ILP, int OP> void run_chain(const char* name, int waves_per_simd, double* d) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves -> one per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((fma_chain<ILP, OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((fma_chain<ILP, OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_wave = (double)iters * 16 * ILP;
    const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-8s ILP %2d waves/SIMD %d : %8.3f ms  %6.3f ns per wave-instr per SIMD (= %5.2f cyc @2.4GHz)  %6.1f T-instr-lanes/s\n", name, ILP, waves_per_simd, ms,
           ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4, instr_per_wave * waves_per_simd * 1024 * 64 / (ms * 1e-3) / 1e12);
}

__global__ void __launch_bounds__(256) fill16(double2* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = make_double2((double)i, 1.0);
}
__global__ void __launch_bounds__(256) fill16_nt(double2* __restrict__ p, long long n) {
    typedef double v2 __attribute__((ext_vector_type(2)));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { v2 v = {(double)i, 1.0}; __builtin_nontemporal_store(v, (v2*)p + i); }
}
__global__ void __launch_bounds__(256) read16(const double2* __restrict__ p, long long n, double* out) {
    double s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
__global__ void empty_kernel() {}
template <int NT> __global__ void __launch_bounds__(NT) fill8(double* __restrict__ p, long long n) {
    const long long i = (long long)blockIdx.x * NT + threadIdx.x; if (i < n) p[i] = (double)i;
}
template <int NT> __global__ void __launch_bounds__(NT) fill16w(double2* __restrict__ p, long long n) {
    const long long i = (long long)blockIdx.x * NT + threadIdx.x; if (i < n) p[i] = make_double2((double)i, 1.0);
}
// each 64-thread WG writes `per` consecutive 16-B units per lane-slot (a 64*16*per byte contiguous chunk): models a wave storing a tile
template <int PER> __global__ void __launch_bounds__(64) fill16_chunk(double2* __restrict__ p, long long n) {
    const long long base = (long long)blockIdx.x * 64 * PER;
#pragma unroll
    for (int r = 0; r < PER; ++r) { const long long i = base + r * 64 + threadIdx.x; if (i < n) p[i] = make_double2((double)i, 1.0); }
}
// persistent grid, chunks of 4 KB handed out by an atomic counter
__global__ void __launch_bounds__(256) fill16_dyn(double2* __restrict__ p, long long n, unsigned* ctr) {
    __shared__ unsigned c;
    const long long nchunks = (n + 255) / 256;
    for (;;) {
        if (threadIdx.x == 0) c = atomicAdd(ctr, 1u);
        __syncthreads();
        const long long ch = c;
        __syncthreads();
        if (ch >= nchunks) break;
        const long long i = ch * 256 + threadIdx.x; if (i < n) p[i] = make_double2((double)i, 1.0);
    }
}
// one-shot grid, but block b writes the 4 KB chunk number perm(b): NS interleaved sequential streams (stream = b % NS)
__global__ void __launch_bounds__(256) fill16_streams(double2* __restrict__ p, long long n, int NS) {
    const long long nchunks = n / 256;
    const long long per = nchunks / NS;                    // chunks per stream
    const long long b = blockIdx.x;
    const long long chunk = (b % NS) * per + (b / NS);
    if (b / NS >= per) return;
    p[chunk * 256 + threadIdx.x] = make_double2((double)chunk, 1.0);
}
// persistent static grid-stride fill that records each block's finish time
__global__ void __launch_bounds__(256) fill16_stamp(double2* __restrict__ p, long long n, long long* stamps) {
    const long long t0 = wall_clock64();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = make_double2((double)i, 1.0);
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}
__global__ void __launch_bounds__(256) read8(const double* __restrict__ p, long long n, double* out) {
    double s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "all";
    double* d; CK(hipMalloc(&d, 1024));
    if (!strcmp(mode, "all") || !strcmp(mode, "valu")) {
        for (int w : {1, 2, 4}) {
            if (w == 1) { run_chain<1, 0>("fma", w, d); run_chain<2, 0>("fma", w, d); run_chain<4, 0>("fma", w, d); run_chain<8, 0>("fma", w, d); run_chain<16, 0>("fma", w, d); }
            else { run_chain<1, 0>("fma", w, d); run_chain<4, 0>("fma", w, d); run_chain<8, 0>("fma", w, d); }
        }
        run_chain<8, 1>("mul", 1, d); run_chain<8, 2>("add", 1, d); run_chain<1, 1>("mul", 1, d); run_chain<1, 2>("add", 1, d);
    }
    if (!strcmp(mode, "all") || !strcmp(mode, "mem") || !strcmp(mode, "calib")) {
        // streaming stores / loads over 1 GiB (4 x the Infinity Cache) and over 136 MB x 4 rotating buffers, plus launch overhead
        const long long big = 1LL << 30;
        double2* buf; CK(hipMalloc(&buf, big));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time_it = [&](const char* name, auto launch, double bytes, int reps) {
            for (int i = 0; i < 2; ++i) launch(i);
            CK(hipDeviceSynchronize());
            std::vector<float> ms(reps);
            for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); launch(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[i], e0, e1)); }
            std::sort(ms.begin(), ms.end());
            // back-to-back: one event pair around `reps` launches
            CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float tot; CK(hipEventElapsedTime(&tot, e0, e1));
            printf("%-28s median %8.2f us  min %8.2f us  back-to-back avg %8.2f us   %7.0f GB/s (median) %7.0f GB/s (b2b)\n", name, ms[reps / 2] * 1e3, ms[0] * 1e3, tot / reps * 1e3,
                   bytes / (ms[reps / 2] * 1e-3) / 1e9, bytes / (tot / reps * 1e-3) / 1e9);
        };
        time_it("empty kernel", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); }, 0, 50);
        time_it("fill16 1GiB grid=n/256", [&](int) { hipLaunchKernelGGL(fill16, dim3((unsigned)(big / 16 / 256)), dim3(256), 0, 0, buf, big / 16); }, (double)big, 10);
        time_it("fill16 1GiB grid=2048", [&](int) { hipLaunchKernelGGL(fill16, dim3(2048), dim3(256), 0, 0, buf, big / 16); }, (double)big, 10);
        time_it("fill16_nt 1GiB grid=n/256", [&](int) { hipLaunchKernelGGL(fill16_nt, dim3((unsigned)(big / 16 / 256)), dim3(256), 0, 0, buf, big / 16); }, (double)big, 10);
        time_it("read16 1GiB grid=n/256/4", [&](int) { hipLaunchKernelGGL(read16, dim3((unsigned)(big / 16 / 256 / 4)), dim3(256), 0, 0, buf, big / 16, d); }, (double)big, 10);
        const long long jb = 65536LL * (247 + 13) * 8;   // J + Y of the flagship workload, 141.8 MB
        time_it("fill16 141.8MB x4 rot grid=n/256", [&](int i) { hipLaunchKernelGGL(fill16, dim3((unsigned)(jb / 16 / 256)), dim3(256), 0, 0, buf + (i % 4) * (jb / 16), jb / 16); }, (double)jb, 40);
        time_it("fill16_nt 141.8MB x4 rot", [&](int i) { hipLaunchKernelGGL(fill16_nt, dim3((unsigned)(jb / 16 / 256)), dim3(256), 0, 0, buf + (i % 4) * (jb / 16), jb / 16); }, (double)jb, 40);
        time_it("fill16 141.8MB x4 rot grid=2048", [&](int i) { hipLaunchKernelGGL(fill16, dim3(2048), dim3(256), 0, 0, buf + (i % 4) * (jb / 16), jb / 16); }, (double)jb, 40);
        unsigned* ctr; CK(hipMalloc(&ctr, 4));
        double2* b4[4]; for (int i = 0; i < 4; ++i) b4[i] = buf + i * (jb / 16);
        time_it("fill8  141.8MB x4 rot WG=256", [&](int i) { hipLaunchKernelGGL(fill8<256>, dim3((unsigned)(jb / 8 / 256)), dim3(256), 0, 0, (double*)b4[i % 4], jb / 8); }, (double)jb, 40);
        time_it("fill8  141.8MB x4 rot WG=64", [&](int i) { hipLaunchKernelGGL(fill8<64>, dim3((unsigned)(jb / 8 / 64)), dim3(64), 0, 0, (double*)b4[i % 4], jb / 8); }, (double)jb, 40);
        time_it("fill16 141.8MB x4 rot WG=64", [&](int i) { hipLaunchKernelGGL(fill16w<64>, dim3((unsigned)(jb / 16 / 64)), dim3(64), 0, 0, b4[i % 4], jb / 16); }, (double)jb, 40);
        time_it("fill16_chunk<20> (20KB/wave) x4 rot", [&](int i) { hipLaunchKernelGGL(fill16_chunk<20>, dim3((unsigned)((jb / 16 + 64 * 20 - 1) / (64 * 20))), dim3(64), 0, 0, b4[i % 4], jb / 16); }, (double)jb, 40);
        time_it("fill16_chunk<8> (8KB/wave) x4 rot", [&](int i) { hipLaunchKernelGGL(fill16_chunk<8>, dim3((unsigned)((jb / 16 + 64 * 8 - 1) / (64 * 8))), dim3(64), 0, 0, b4[i % 4], jb / 16); }, (double)jb, 40);
        time_it("fill16_dyn persistent 2048 WG", [&](int i) { CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL(fill16_dyn, dim3(2048), dim3(256), 0, 0, b4[i % 4], jb / 16, ctr); }, (double)jb, 40);
        for (int ns : {1, 8, 64, 256, 1024, 4096}) {
            char nm[64]; snprintf(nm, 64, "fill16_streams NS=%d x4 rot", ns);
            time_it(nm, [&](int i) { hipLaunchKernelGGL(fill16_streams, dim3((unsigned)(jb / 16 / 256)), dim3(256), 0, 0, b4[i % 4], jb / 16, ns); }, (double)jb, 40);
        }
        {
            long long* st; CK(hipMalloc(&st, 2048 * 16));
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(fill16_stamp, dim3(2048), dim3(256), 0, 0, b4[rep % 4], jb / 16, st);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(4096); CK(hipMemcpy(h.data(), st, 2048 * 16, hipMemcpyDeviceToHost));
            long long tmin = h[0]; for (int i = 0; i < 2048; ++i) tmin = std::min(tmin, h[2 * i]);
            std::vector<double> fin(2048), sta(2048); for (int i = 0; i < 2048; ++i) { fin[i] = (h[2 * i + 1] - tmin) / 100.0; sta[i] = (h[2 * i] - tmin) / 100.0; }   // wall_clock64 = 100 MHz
            std::vector<double> f2 = fin; std::sort(f2.begin(), f2.end()); std::vector<double> s2 = sta; std::sort(s2.begin(), s2.end());
            printf("fill16_stamp persistent 2048 blocks: start us min %.2f med %.2f max %.2f | finish us min %.2f p10 %.2f med %.2f p90 %.2f max %.2f\n", s2[0], s2[1024], s2[2047], f2[0], f2[204], f2[1024], f2[1843], f2[2047]);
            // by XCD (block % 8)
            for (int x = 0; x < 8; ++x) { double m = 0, mx = 0; int c = 0; for (int i = x; i < 2048; i += 8) { m += fin[i]; mx = std::max(mx, fin[i]); ++c; } printf("   xcd %d mean finish %.2f max %.2f\n", x, m / c, mx); }
        }
        time_it("read8 1GiB", [&](int) { hipLaunchKernelGGL(read8, dim3(8192), dim3(256), 0, 0, (const double*)buf, big / 8, d); }, (double)big, 10);
        time_it("fill16 141.8MB same buffer", [&](int) { hipLaunchKernelGGL(fill16, dim3((unsigned)(jb / 16 / 256)), dim3(256), 0, 0, buf, jb / 16); }, (double)jb, 40);
    }
    return 0;
}
