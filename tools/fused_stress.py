#!/usr/bin/env python3
"""tools/fused_stress.py [launches] -- long-running stress of the fused rollout -> Jacobians launch's consumer side.

The Jacobian workgroups of a fused launch read states that OTHER workgroups of the same launch wrote (csrc/fused.hpp: relaxed poll of a progress word, a
compiler-only barrier, then agent-scope loads; relies on in-order load issue per wave and on `sc1` loads being served from memory across XCDs).  The tests
exercise that for a few hundred launches; this runs it for `launches` (default 100 000) over the shapes whose tiles SHARE 128-byte lines (M = 66 / 130 /
4 094 trajectories, K = 37 / 101 knots), fp32 and fp64, both layouts, with 3 .. 192 Jacobian workgroups (RDHIP_TUNE fused_jac_workgroups), and checks
EVERY launch on the device:
  * the simulated states equal rdhip_rollout's bit for bit in the one-launch form (to rounding, and bit for bit from launch to launch, where the call runs as
    two launches or in pipelined pieces),
  * the Jacobians equal the FIRST launch's bit for bit (a stale read of any state changes bits) -- and that first result is held to rdhip_jacobian's on
    the simulated states at 1e-14 (fp64) / 2e-6 (fp32) relative, the tolerance of tests/test_gpu_round4.py.
Before every launch the states of knots 2 .. K are overwritten with NaN, so a consumer that reads a state too early cannot find last launch's value.
Prints one JSON line per (shape, dtype, layout) and a summary line; exit code 1 on the first mismatch."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rdhip as RD  # noqa: E402
from bench_configs import rigid_inputs  # noqa: E402

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
SHAPES = [(66, 37), (130, 37), (66, 101), (130, 101), (4094, 37), (4094, 101), (4096, 37)]      # (the last: fp32 time-major takes the two-workgroups-per-CU kernel)
NJS = [3, 16, 64, 192, 0]          # Jacobian workgroups (0: every CU the rollout groups leave)
sig, AD = RD.StaticReturn(), RD.ForwardAD()


def run(M, K, dtype, tm, launches, rng):
    model = RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0]))
    fun = RD.DiscretizedDynamics(model, RD.RK4)
    n, m = 13, 6
    T = torch.float64 if dtype == np.float64 else torch.float32
    Zn = rigid_inputs(M * K, m, rng).astype(dtype).reshape(M, K, n + m)
    lay = (lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))) if tm else (lambda a: np.ascontiguousarray(a))
    tk = np.tile(np.arange(K) * 0.02, (M, 1)); dk = np.full((M, K), 0.02)
    init = torch.as_tensor(lay(Zn)).cuda()
    if tm:
        init[1:, :, :n] = float("nan")
    else:
        init[:, 1:, :n] = float("nan")
    st = {"data": init.clone(), "t": torch.as_tensor(lay(tk)).cuda(), "dt": torch.as_tensor(lay(dk)).cuda()}
    RD.rollout_(sig, fun, st, time_major=tm)                 # (64 groups and fewer: the fused kernel's own rollout role -- the same waves, the same bits)
    ref_states = st["data"].clone()
    flat = RD.PackedTrajectory(n, m, ref_states.reshape(M * K, n + m), st["t"].reshape(-1), st["dt"].reshape(-1))
    Jref = RD.alloc_jacobians(fun, M * K, init)
    RD.jacobian_(sig, AD, fun, Jref, None, flat)
    J = RD.alloc_jacobians(fun, M * K, init)
    first, fused, done, t0 = None, 0, 0, time.time()
    for it in range(launches):
        seg = max(1, launches // len(NJS))
        if it % seg == 0:
            RD.tune(f"fused_jac_workgroups={NJS[min(it // seg, len(NJS) - 1)]}")
        st["data"].copy_(init)
        J.fill_(float("nan"))
        k = RD.rollout_jacobian_(sig, AD, fun, J, st, None, time_major=tm)
        fused += 1 if k == 1 else 0
        if k == 1 and not torch.equal(st["data"], ref_states):            # the one-launch form: the same rollout waves as rdhip_rollout, the same bits
            return {"M": M, "K": K, "dtype": str(np.dtype(dtype)), "time_major": tm, "launch": it, "error": "states differ from rdhip_rollout",
                    "max_diff": float((st["data"] - ref_states).abs().nan_to_num(nan=1e30).max())}
        if first is None:
            first_states = st["data"].clone()
            # (two launches / pipelined pieces: a piece boundary moves steps between the cascade kernel's chunk loop and its tail, two compilations of the
            # same step -- rounding-level differences from rdhip_rollout, identical from launch to launch)
            if not torch.allclose(first_states, ref_states, rtol=1e-12 if dtype == np.float64 else 1e-5, atol=1e-13 if dtype == np.float64 else 1e-6, equal_nan=False):
                return {"M": M, "K": K, "dtype": str(np.dtype(dtype)), "time_major": tm, "launch": it, "error": "states differ from rdhip_rollout beyond rounding"}
            first = J.clone()
            rel = float(((first - Jref).abs().max() / Jref.abs().max()).item())
            if not rel <= (1e-14 if dtype == np.float64 else 2e-6):
                return {"M": M, "K": K, "dtype": str(np.dtype(dtype)), "time_major": tm, "launch": it, "error": "Jacobians differ from rdhip_jacobian", "rel": rel}
        elif not torch.equal(st["data"], first_states):
            return {"M": M, "K": K, "dtype": str(np.dtype(dtype)), "time_major": tm, "launch": it, "error": "states differ from the first launch"}
        elif not torch.equal(J, first):
            bad = (J != first) | torch.isnan(J)
            return {"M": M, "K": K, "dtype": str(np.dtype(dtype)), "time_major": tm, "launch": it, "error": "Jacobians differ from the first launch", "entries": int(bad.sum())}
        done += 1
    RD.tune("reset")
    return {"M": M, "K": K, "dtype": str(np.dtype(dtype)), "time_major": tm, "launches": done, "one_launch_form": fused, "seconds": round(time.time() - t0, 1), "ok": True}


def main():
    rng = np.random.default_rng(2026)
    cases = [(M, K, dt, tm) for (M, K) in SHAPES for dt in (np.float64, np.float32) for tm in (True, False)]
    # weight the launch budget towards the small shapes (a 4 094 x 101 launch costs 40x a 66 x 37 one)
    w = np.array([1.0 / (M * K) ** 0.5 for M, K, _, _ in cases]); w = w / w.sum()
    total, fused, bad = 0, 0, 0
    for (M, K, dt, tm), wi in zip(cases, w):
        r = run(M, K, dt, tm, max(50, int(TOTAL * wi)), rng)
        print(json.dumps(r), flush=True)
        if not r.get("ok"):
            bad += 1
            break
        total += r["launches"]; fused += r["one_launch_form"]
    print(json.dumps({"summary": True, "launches": total, "in_the_one_launch_form": fused, "mismatches": bad}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
