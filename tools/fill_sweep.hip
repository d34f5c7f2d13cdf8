// fill_sweep.hip -- developer tool: streaming-fill bandwidth (16 B per lane, non-temporal, persistent grid) as a function of the footprint, to
// separate what the memory system does at multi-GB footprints from what the Jacobian kernels do.  hipcc --offload-arch=gfx950 -O3 tools/fill_sweep.hip -o tools/fill_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) fill(v2* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { v2 v = {(double)i, 1.0}; __builtin_nontemporal_store(v, p + i); }
}
// tile-strided variant: the Jacobian kernels' order -- workgroup b writes 126 KB slabs b, b + G, b + 2G, ...
__global__ void __launch_bounds__(256) fill_tiles(v2* __restrict__ p, long long ntiles) {
    const int UNITS = 64 * 247 * 8 / 16;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x)
        for (int u = threadIdx.x; u < UNITS; u += 256) { v2 v = {(double)u, 1.0}; __builtin_nontemporal_store(v, p + t * UNITS + u); }
}
int main() {
    const double GB[] = {0.147, 0.59, 1.18, 2.36, 4.7, 9.4};
    for (double gb : GB) {
        const long long n = (long long)(gb * 1e9 / 16);
        v2* buf[2]; CK(hipMalloc(&buf[0], n * 16)); CK(hipMalloc(&buf[1], n * 16));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<float> ms(20);
            for (int i = 0; i < 23; ++i) {
                CK(hipEventRecord(e0));
                if (mode == 0) fill<<<2048, 256>>>(buf[i & 1], n); else fill_tiles<<<256, 256>>>(buf[i & 1], n / (64 * 247 * 8 / 16));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); if (i >= 3) ms[i - 3] = t;
            }
            std::sort(ms.begin(), ms.end());
            printf("%-12s footprint %6.2f GB x2 buffers: median %8.1f us (%5.0f GB/s)  min %8.1f us (%5.0f GB/s)\n", mode ? "tile-strided" : "linear", gb, ms[10] * 1e3, gb / (ms[10] * 1e-3), ms[0] * 1e3, gb / (ms[0] * 1e-3));
        }
        CK(hipFree(buf[0])); CK(hipFree(buf[1]));
    }
    return 0;
}
