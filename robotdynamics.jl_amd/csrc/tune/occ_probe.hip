#include <cstdio>
#include "harness_tune.hpp"
#include "../fam_lie.hpp"
using namespace rd;
int main() {
    {
        using Fn = Fun<Cartpole<double>, RULE_RK4>;
        auto k = jac_ad_small_kernel<Fn, double>;
        constexpr size_t lds = SmallLds<Fn, double>::BYTES; constexpr int W = SmallLds<Fn, double>::WAVES;
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k, W * 64, lds);
        printf("cartpole small: lds %zu threads %d -> %d (%s)\n", lds, W * 64, n, hipGetErrorString(e));
    }
    {
        using Fn = Fun<LieSatellite<double>, RULE_RK4>;
        auto k = jac_ad_split_kernel<Fn, double, LieSatPlan, 4>;
        constexpr size_t lds = SplitLds<Fn, double, LieSatPlan, 4>::BYTES;
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k, 512, lds);
        printf("liesat split: lds %zu threads 512 -> %d (%s)\n", lds, n, hipGetErrorString(e));
    }
    return 0;
}
