#!/usr/bin/env python3
"""tools/mixed_stress.py [launches] -- stress of the ONE-launch mixed-model call (csrc/mixed.hpp: a satellite's and a cartpole's rollouts and all their Jacobians in
one grid; BASELINE config 5's path).  Inside that launch the cartpole's Jacobian waves read states that OTHER workgroups of the same launch wrote (plain stores,
one agent-scope release fence per rollout wave, a flag word raised by the wave that counts last), and the satellite's tiles are consumed through progress words
as in the fused launch (tools/fused_stress.py).  Shapes whose groups share 128-byte lines, ragged last groups, both layouts, fp32 and fp64, both placements of
the cartpole's rollout (idle waves / workgroups of its own), the two-workgroups-per-CU form; the states of knots 2 .. K are overwritten with NaN before
every launch.  Checked on the device after EVERY launch: all four outputs equal the first launch's bit for bit; the first launch's states equal
rdhip_rollout's (satellite: bit for bit, cartpole: to rounding) and its Jacobians rdhip_jacobian's on those states to 2e-6 / 1e-13 relative."""
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

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
SHAPES = [(68, 70, 37), (132, 200, 37), (132, 64, 101), (2052, 2049, 21), (4096, 4100, 9)]      # (satellite counts: multiples of 4, what the fp32 time-major form wants)
sig, AD = RD.StaticReturn(), RD.ForwardAD()


def stack(model, n, m, M, K, dtype, tm, rng):
    Zn = (rigid_inputs(M * K, m, rng) if n == 13 else rng.random((M * K, n + m)) - 0.3).astype(dtype).reshape(M, K, n + m)
    lay = (lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))) if tm else np.ascontiguousarray
    init = torch.as_tensor(lay(Zn)).cuda()
    if tm:
        init[1:, :, :n] = float("nan")
    else:
        init[:, 1:, :n] = float("nan")
    tk = np.tile(np.arange(K) * 0.02, (M, 1)); dk = np.full((M, K), 0.02)
    return init, torch.as_tensor(lay(tk)).cuda(), torch.as_tensor(lay(dk)).cuda()


def run(MA, MB, K, dtype, tm, idle, launches, rng):
    sat, cart = RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0])), RD.Cartpole()
    funs = [RD.DiscretizedDynamics(sat, RD.RK4), RD.DiscretizedDynamics(cart, RD.RK4)]
    inits, sts, Js = [], [], []
    for (n, m, M) in ((13, 6, MA), (4, 1, MB)):
        init, t, dt = stack(None, n, m, M, K, dtype, tm, rng)
        inits.append(init); sts.append({"data": init.clone(), "t": t, "dt": dt})
    Js = [RD.alloc_jacobians(funs[i], (MA, MB)[i] * K, sts[i]["data"]) for i in range(2)]
    tag = {"MA": MA, "MB": MB, "K": K, "dtype": str(np.dtype(dtype)), "time_major": tm, "mixed_idle": idle}
    # references: the separate calls
    refs = []
    for i in range(2):
        s = {"data": inits[i].clone(), "t": sts[i]["t"], "dt": sts[i]["dt"]}
        RD.rollout_(sig, funs[i], s, time_major=tm)
        M = (MA, MB)[i]; nz = (19, 5)[i]
        flat = RD.PackedTrajectory((13, 4)[i], (6, 1)[i], s["data"].reshape(M * K, nz), s["t"].reshape(-1), s["dt"].reshape(-1))
        Jr = RD.alloc_jacobians(funs[i], M * K, s["data"])
        RD.jacobian_(sig, AD, funs[i], Jr, None, flat)
        refs.append((s["data"], Jr))
    RD.tune(f"mixed_idle={idle}")
    try:
        call = RD.PreparedRolloutJacobianMulti(sig, AD, funs, Js, sts, None, time_major=tm)
        first, t0, kernel = None, time.time(), ""
        for it in range(launches):
            for i in range(2):
                sts[i]["data"].copy_(inits[i]); Js[i].fill_(float("nan"))
            call()
            if call.launches != 1:
                return dict(tag, error=f"{call.launches} launches: not the one-launch form", kernel=RD.last_kernel_name()[:80])
            if first is None:
                kernel = RD.last_kernel_name()
                first = [sts[0]["data"].clone(), sts[1]["data"].clone(), Js[0].clone(), Js[1].clone()]
                if not torch.equal(first[0], refs[0][0]):
                    return dict(tag, error="the satellite's states differ from rdhip_rollout's")
                if not torch.allclose(first[1], refs[1][0], rtol=1e-5 if dtype == np.float32 else 1e-12, atol=1e-6 if dtype == np.float32 else 1e-13):
                    return dict(tag, error="the cartpole's states differ from rdhip_rollout's beyond rounding")
                for i in range(2):
                    rel = float(((first[2 + i] - refs[i][1]).abs().max() / refs[i][1].abs().max()).item())
                    if not rel <= (2e-6 if dtype == np.float32 else 1e-13):
                        return dict(tag, error=f"model {i}: Jacobians differ from rdhip_jacobian's", rel=rel)
            else:
                got = [sts[0]["data"], sts[1]["data"], Js[0], Js[1]]
                for i in range(4):
                    if not torch.equal(got[i], first[i]):
                        bad = (got[i] != first[i]) | torch.isnan(got[i])
                        return dict(tag, launch=it, error=f"output {i} differs from the first launch", entries=int(bad.sum()))
    finally:
        RD.tune("reset")
    return dict(tag, launches=launches, seconds=round(time.time() - t0, 1), two_per_cu="68" in kernel.split("Cascade")[-1][:60] if kernel else None, ok=True)


def main():
    rng = np.random.default_rng(2027)
    cases = [(MA, MB, K, dt, tm, idle) for (MA, MB, K) in SHAPES for dt in (np.float32, np.float64) for tm in (True, False) for idle in (0, 1)]
    w = np.array([1.0 / ((MA + MB) * K) ** 0.5 for MA, MB, K, *_ in cases]); w = w / w.sum()
    total, bad = 0, 0
    for (MA, MB, K, dt, tm, idle), wi in zip(cases, w):
        r = run(MA, MB, K, dt, tm, idle, max(30, int(TOTAL * wi)), rng)
        print(json.dumps(r), flush=True)
        if not r.get("ok"):
            bad += 1
            break
        total += r["launches"]
    print(json.dumps({"summary": True, "launches": total, "mismatches": bad}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
