// occ_place.hip -- developer tool: how many workgroups of a given shape (threads, registers, LDS) does the dispatcher REALLY place on a CU at once?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor counts wave slots per SIMD; whether a second workgroup's waves find a legal placement is another matter.)
// Every workgroup stamps its start, spins 30 us and stamps its end; a grid of 2 x 256 (or 3 x 256) either starts together or in rounds.
// Build: hipcc --offload-arch=gfx950 -O3 tune/occ_place.hip -o tune/bin/occ_place
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int THREADS, int VREG, int MINW>
__global__ void __launch_bounds__(THREADS, MINW) probe(unsigned long long* out, int spin_ticks) {
    extern __shared__ unsigned char smem[];
    if constexpr (VREG >= 255) asm volatile("v_mov_b32 v254, 0" ::: "v254");
    else if constexpr (VREG >= 168 && VREG < 255) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    else if constexpr (VREG >= 255) asm volatile("v_mov_b32 v254, 0" ::: "v254");
    else if constexpr (VREG >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    else if constexpr (VREG >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    const unsigned long long t0 = __builtin_readcyclecounter() * 0 + wall_clock64();
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = wall_clock64(); }
}
template <int THREADS, int VREG, int MINW> int run(const char* name, size_t lds, int grid, unsigned long long* d) {
    auto k = probe<THREADS, VREG, MINW>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = -1; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k, THREADS, lds));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)k));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, d, 3000); CK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(2 * grid); CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ULL; for (int b = 0; b < grid; ++b) t0 = std::min(t0, h[2 * b]);
    int early = 0; double last = 0; for (int b = 0; b < grid; ++b) { const double s = (h[2 * b] - t0) * 0.01; if (s < 10) ++early; last = std::max(last, s); }
    printf("%-34s threads %d regs %d lds %zu: API occupancy %d; of %d workgroups %d started within 10 us, the last at %.1f us\n", name, THREADS, fa.numRegs, lds, occ, grid, early, last);
    return 0;
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 4096 * 16));
    run<384, 168, 3>("6 waves, 168 regs, 78 KB", 78016, 512, d);
    run<384, 128, 4>("6 waves, 128 regs, 78 KB", 78016, 512, d);
    run<384, 168, 3>("6 waves, 168 regs, 40 KB", 40000, 512, d);
    run<384, 168, 3>("6 waves, 168 regs, 8 KB", 8192, 512, d);
    run<512, 128, 4>("8 waves, 128 regs, 78 KB", 78016, 512, d);
    run<512, 168, 3>("8 waves, 168 regs, 8 KB", 8192, 512, d);
    run<256, 168, 3>("4 waves, 168 regs, 78 KB", 78016, 512, d);
    run<256, 168, 3>("4 waves, 168 regs, 50 KB", 51200, 768, d);
    run<256, 255, 2>("4 waves, 256 regs, 78 KB", 78016, 512, d);
    run<384, 96, 5>("6 waves, 96 regs, 78 KB", 78016, 512, d);
    run<384, 168, 3>("6 waves, 168 regs, 76 KB", 76 * 1024, 512, d);
    run<384, 168, 3>("6 waves, 168 regs, 64 KB", 64 * 1024, 512, d);
    return 0;
}
