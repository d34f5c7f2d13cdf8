#!/usr/bin/env python3
"""tools/c5_overlap.py [M] [reps] -- BASELINE config 5 as ONE rdhip_rollout_jacobian_multi call on a context that owns its stream with launch overlap on
(rdhip_ctx_set_overlap): a repeated call neither forks nor joins (the second model lives on the context's internal stream), so the context's stream
carries the long model's launches back to back.  Prints the time per call by the library's stopwatch, beside the same call on a plain context."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rdhip as RD  # noqa: E402
from rdhip import _lib as L  # noqa: E402
import bench_configs as BC  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
K = 256
rng = np.random.default_rng(7)


def build(ctx):
    funs, stacks, Js = [], [], []
    for model, mctl in ((RD.Cartpole(), 1), (RD.Satellite(), 6)):
        n = RD.state_dim(model)
        Zn = (BC.rigid_inputs(M * K, mctl, rng) if n == 13 else rng.random((M * K, n + mctl))).astype(np.float32)
        fun = RD.DiscretizedDynamics(model, RD.RK4)
        tk = np.arange(K) * 0.01
        st = {"data": torch.from_numpy(Zn.reshape(K, M, n + mctl)).to("cuda:0"), "t": torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[:, None], (K, M)))).to("cuda:0"),
              "dt": torch.full((K, M), 0.01, dtype=torch.float64, device="cuda:0")}
        funs.append(fun); stacks.append(st); Js.append(RD.alloc_jacobians(fun, M * K, st["data"]))
    return RD.PreparedRolloutJacobianMulti(RD.StaticReturn(), RD.ForwardAD(), funs, Js, stacks, None, time_major=True, ctx=ctx), (funs, stacks, Js)


# (one form per process: HIP maps a process's streams onto four hardware queues, and the plain form's child streams stay alive with the default context)
if len(sys.argv) > 3 and sys.argv[3] == "plain":
    plain, keep0 = build(None)
    print(f"{2 * M} trajectories x {K} knots in one call: plain context {BC.timed(lambda i: plain(), reps, warm=3):.1f} us")
    sys.exit(0)
ctx = L.Context(0)
ctx.set_overlap(True)
call, keep = build(ctx)
us_ov = BC.timed_ctx(lambda i: call(), reps, ctx, warm=3)
print(f"{2 * M} trajectories x {K} knots in one call: own stream + launch overlap {us_ov:.1f} us")
ctx.close()
