#!/usr/bin/env python3
"""tools/c5_one_call.py [M] [reps] [tj] -- BASELINE config 5 as ONE rdhip_rollout_jacobian_multi call (M Cartpole + M Satellite trajectories x 256 knots, fp32; time-major,
or with a third argument `tj` trajectory-major: the reference's own layout, a vector of trajectories), issued `reps` times back to back; prints the HIP-event time per call.
The command tools/c5_timeline.sh traces."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rdhip as RD  # noqa: E402
import bench_configs as BC  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
K = 256
TM = not (len(sys.argv) > 3 and sys.argv[3] == "tj")
rng = np.random.default_rng(7)
funs, stacks, Js = [], [], []
for model, mctl in ((RD.Cartpole(), 1), (RD.Satellite(), 6)):
    n = RD.state_dim(model)
    Zn = (BC.rigid_inputs(M * K, mctl, rng) if n == 13 else rng.random((M * K, n + mctl))).astype(np.float32)
    fun = RD.DiscretizedDynamics(model, RD.RK4)
    tk = np.arange(K) * 0.01
    if TM:
        st = {"data": torch.from_numpy(Zn.reshape(K, M, n + mctl)).to("cuda:0"), "t": torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[:, None], (K, M)))).to("cuda:0"),
              "dt": torch.full((K, M), 0.01, dtype=torch.float64, device="cuda:0")}
    else:
        st = {"data": torch.from_numpy(Zn.reshape(M, K, n + mctl)).to("cuda:0"), "t": torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[None, :], (M, K)))).to("cuda:0"),
              "dt": torch.full((M, K), 0.01, dtype=torch.float64, device="cuda:0")}
    funs.append(fun); stacks.append(st); Js.append(RD.alloc_jacobians(fun, M * K, st["data"]))
call = RD.PreparedRolloutJacobianMulti(RD.StaticReturn(), RD.ForwardAD(), funs, Js, stacks, None, time_major=TM)
us = BC.timed(lambda i: call(), reps, warm=3)
print(f"{2 * M} trajectories x {K} knots in one call ({'time-major' if TM else 'trajectory-major'}, {call.launches} launch(es) on the context's stream): {us:.1f} us")
