// tune_default.hpp -- the tuning knobs of the kernels as ONE compile-time policy.  The library (and the hipRTC plug-in) is built with rd::DefaultTune and
// nothing else; the developer harnesses under csrc/tune/ derive a policy from it (tune/harness_tune.hpp turns their -DRD_* switches into members and
// owns the device globals the stamps write to) and name it in RD_TUNE before kernels.hpp is included.  Product code reads `Tune::X` inside
// `if constexpr`: a knob that is off costs no instruction and no register, and the hot loops carry no preprocessor branches.
#pragma once
#include "scalar.hpp"

namespace rd {

struct DefaultTune {
    static constexpr int ABLATE = 0;               // 2: no arithmetic in the ForwardAD columns, same store pattern; 3: no tile stores (what does the arithmetic alone cost?)
    static constexpr bool PLAIN_STORES = false;    // tile stores without the non-temporal hint (measured 3-5 % slower)
    static constexpr bool NO_EARLY = false;        // no early part of a workgroup's first tile (TablePlanS)
    static constexpr bool NPASS_ANY = false;       // the two-pass copy of two copy waves for fp64 tiles too (tune/tune_ng.hip)
    static constexpr bool MULTI_NO_RING = false;   // batch tables: every wave resolves its own tiles instead of reading the workgroup's LDS table
    static constexpr bool COL_BARRIER = false;     // a scheduling barrier behind every compile-time ForwardAD column
    static constexpr int TILE_MAP = 1;             // tile -> workgroup order of static launches: 0 plain at every size, 2 XCD ranges at every size, 1 XCD ranges from 64 tiles per workgroup on
    static constexpr bool DYN_F32 = false;         // the dynamic tile schedule for fp32 tiles too (tune/tune_dyn.hip -DTT=float)
    static constexpr int DYN_CHUNK = 1;            // tiles per fetch of the dynamic schedule's counter
    static constexpr int DYN_XR = 128;             // tiles per workgroup from which every XCD range gets a counter of its own
    static constexpr bool HELPER_CHAIN_FIRST = false;   // helper wave: the next tile's chain in front of barrier E (tune/tune_ift.hip)
    static constexpr int F32_MINB = 2;             // register budget of the fp32 ForwardAD tile kernels, in waves per SIMD (see kernels.hpp)
    static constexpr int FUSED_F32_MINW = 2;       // the same for the fp32 fused rollout -> Jacobians kernel (fused.hpp)
    static constexpr bool NO_HELPER_STAGE = false;     // the helper wave does not stage the next tile's inputs (A/B, tune/tune_ift.hip)
    static constexpr bool COPY_PREFETCH_NONE = false;  // no body prefetches through its copy waves (A/B of the models that do, tune/tune_plan.hip)
    static constexpr bool COPY_PREFETCH_ALL = false;   // every split-role body prefetches through its copy waves (kernels.hpp CopyPrefetchOf)
    static constexpr int COPY_PACE = 0;            // s_sleep units behind every tile store of a copy wave (tune/tune_plan.hip: do paced stores let the compute waves' scratch accesses through?)
    static constexpr int COPY_PRIO = 3;            // issue priority of the copy waves
    static constexpr int IFT_ABL = 0;              // implicit-function-theorem tile kernel: 1 = (h/2) A - I for free, 2 = no Newton solve (implicit.hpp, tune/tune_ift.hip)
    static constexpr bool NO_CASCADE_NEWTON = false;    // the full Newton iteration instead of the block-by-block one (implicit.hpp)
    static RD_DEV int tile_map_runtime() { return 0; }   // >= 100: an order chosen at run time (tune/tune_bign.hip; jac_split_loop)
    static RD_DEV void stamp(int, int) {}                // per-wave phase time stamps (tune/tune_jac.hip)
    static RD_DEV void split_stamp(int, int) {}          // split-role loop: [workgroup][iteration < 8][slot < 16] wall-clock stamps (tune/tune_stamp.hip, tune_c5.hip)
    static RD_DEV void fused_stamp(int) {}               // fused rollout -> Jacobians kernel: start / end of a workgroup's roles (tune/tune_c5.hip)
};

}  // namespace rd
