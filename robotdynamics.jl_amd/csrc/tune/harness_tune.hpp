// harness_tune.hpp -- developer harnesses only: turns the -DRD_* switches of a tune/*.hip build into the kernels' tuning policy (tune_default.hpp) and owns
// the device globals the stamps write to.  Include FIRST, before any kernel header.  The library never sees this file.
#pragma once
#include "../tune_default.hpp"

namespace rd {
__device__ long long* g_stamps = nullptr;                    // -DRD_STAMP: [slot][64 events]
__device__ long long* g_split_stamp = nullptr;               // -DRD_SPLIT_STAMP: [workgroup][iteration < 8][slot < 16]
__device__ unsigned long long* g_fused_stamp = nullptr;      // -DRD_FUSED_STAMP: role start / end per workgroup
__device__ int g_tile_map = 0;                               // -DRD_TILE_MAP_RUNTIME
struct HarnessTune : DefaultTune {
#if defined(RD_ABLATE)
    static constexpr int ABLATE = RD_ABLATE;
#endif
#if defined(RD_PLAIN_STORES)
    static constexpr bool PLAIN_STORES = true;
#endif
#if defined(RD_NO_EARLY)
    static constexpr bool NO_EARLY = true;
#endif
#if defined(RD_NPASS_ANY)
    static constexpr bool NPASS_ANY = true;
#endif
#if defined(RD_MULTI_NO_RING)
    static constexpr bool MULTI_NO_RING = true;
#endif
#if defined(RD_COL_BARRIER)
    static constexpr bool COL_BARRIER = true;
#endif
#if defined(RD_TILE_MAP)
    static constexpr int TILE_MAP = RD_TILE_MAP;
#endif
#if defined(RD_DYN_F32)
    static constexpr bool DYN_F32 = true;
#endif
#if defined(RD_DYN_CHUNK)
    static constexpr int DYN_CHUNK = RD_DYN_CHUNK;
#endif
#if defined(RD_DYN_XR)
    static constexpr int DYN_XR = RD_DYN_XR;
#endif
#if defined(RD_HELPER_CHAIN_FIRST)
    static constexpr bool HELPER_CHAIN_FIRST = true;
#endif
#if defined(RD_F32_MINB)
    static constexpr int F32_MINB = RD_F32_MINB;
#endif
#if defined(RD_FUSED_F32_MINW)
    static constexpr int FUSED_F32_MINW = RD_FUSED_F32_MINW;
#endif
#if defined(RD_NO_HELPER_STAGE)
    static constexpr bool NO_HELPER_STAGE = true;
#endif
#if defined(RD_COPY_PREFETCH_NONE)
    static constexpr bool COPY_PREFETCH_NONE = true;
#endif
#if defined(RD_COPY_PREFETCH_ALL)
    static constexpr bool COPY_PREFETCH_ALL = true;
#endif
#if defined(RD_COPY_PACE)
    static constexpr int COPY_PACE = RD_COPY_PACE;
#endif
#if defined(RD_COPY_PRIO)
    static constexpr int COPY_PRIO = RD_COPY_PRIO;
#endif
#if defined(RD_IFT_ABL)
    static constexpr int IFT_ABL = RD_IFT_ABL;
#endif
#if defined(RD_NO_CASCADE_NEWTON)
    static constexpr bool NO_CASCADE_NEWTON = true;
#endif
#if defined(RD_TILE_MAP_RUNTIME)
    static RD_DEV int tile_map_runtime() { return g_tile_map; }
#endif
#if defined(RD_STAMP)
    static RD_DEV void stamp(int slot, int ev) { if ((threadIdx.x & 63) == 0 && g_stamps) g_stamps[slot * 64 + ev] = wall_clock64(); }
#endif
#if defined(RD_SPLIT_STAMP)
    static RD_DEV void split_stamp(int it, int slot) {
        const long long c = (long long)wall_clock64(); long long* sp = g_split_stamp;
        if ((threadIdx.x & 63) == 0 && it < 8 && sp) sp[((long long)blockIdx.x * 8 + it) * 16 + slot] = c;
    }
#endif
#if defined(RD_FUSED_STAMP)
    static RD_DEV void fused_stamp(int slot) { if (threadIdx.x == 0 && g_fused_stamp) g_fused_stamp[slot] = wall_clock64(); }
#endif
};
}  // namespace rd
// the -D names above must not leak into the kernel headers as macros (kernels.hpp uses some of them as identifiers)
#undef RD_F32_MINB
#undef RD_DYN_CHUNK
#undef RD_DYN_XR
#define RD_TUNE rd::HarnessTune
