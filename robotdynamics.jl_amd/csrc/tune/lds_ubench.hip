// lds_ubench.hip -- developer tool: issue cost (cycles per instruction, one wave alone on a CU) of LDS write / read flavours
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(64) k(long long* out, double a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef double __attribute__((address_space(3)))* LD;
    typedef v2d __attribute__((address_space(3)))* LD2;
    LD p = (LD)smem; LD2 p2 = (LD2)smem;
    const int lane = threadIdx.x;
    double v0 = a + lane, v1 = a * lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int it = 0; it < 100; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) { p[(2 * r) * 64 + lane] = v0; p[(2 * r + 1) * 64 + lane] = v1; }              // -> ds_write2st64_b64
            else if (MODE == 1) { v2d w = {v0, v1}; p2[r * 64 + lane] = w; }                               // ds_write_b128, lane stride 16 B
            else if (MODE == 2) { p[r * 64 + lane] = v0; }                                                  // ds_write_b64
            else if (MODE == 3) { p[(r * 64 + lane) * 13] = v0; }                                           // ds_write_b64, lane stride 104 B
            else if (MODE == 4) { v0 += p[(2 * r) * 64 + lane]; v1 += p[(2 * r + 1) * 64 + lane]; }        // ds_read2st64_b64 (dependent use)
            else if (MODE == 5) { ((float __attribute__((address_space(3)))*)smem)[r * 64 + lane] = (float)v0; }   // ds_write_b32
            asm volatile("" ::: "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = (long long)(v0 + v1); }
}
int main() {
    long long* d; CK(hipMalloc(&d, 64));
    const char* names[] = {"ds_write2st64_b64 (16 B/lane)", "ds_write_b128 (16 B/lane)", "ds_write_b64 stride 8 B", "ds_write_b64 stride 104 B", "ds_read2st64_b64 + add", "ds_write_b32"};
    for (int m = 0; m < 6; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (m) {
                case 0: k<0><<<1, 64, 65536>>>(d, 1.0); break; case 1: k<1><<<1, 64, 65536>>>(d, 1.0); break; case 2: k<2><<<1, 64, 65536>>>(d, 1.0); break;
                case 3: k<3><<<1, 64, 65536>>>(d, 1.0); break; case 4: k<4><<<1, 64, 65536>>>(d, 1.0); break; default: k<5><<<1, 64, 65536>>>(d, 1.0); break;
            }
            CK(hipDeviceSynchronize());
        }
        long long h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("%-32s %7.1f cycles per instruction\n", names[m], h[0] / 800.0);
    }
    return 0;
}
