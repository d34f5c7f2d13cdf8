#!/usr/bin/env python3
"""tools/rollout_probe.py -- rollout kernel timing over batch shapes (trajectories x knots) and dtypes: microseconds per launch and per step.
RDHIP_TUNE=rollout=plain selects the one-lane-per-trajectory kernel for rigid bodies (default: the cascade kernel)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rdhip as RD  # noqa: E402
from tools.bench_configs import rigid_inputs, timed  # noqa: E402

DEV = "cuda:0"
rng = np.random.default_rng(3)
sat, quad = RD.Satellite(), RD.Quadrotor()
for model, name, m in ((sat, "satellite", 6), (quad, "quadrotor", 4)):
    for dtype in (np.float64, np.float32):
        for M, K in ((64, 256), (512, 256), (2048, 128), (4096, 101), (16384, 64)):
            Zn = rigid_inputs(M * K, m, rng).astype(dtype)
            data = torch.from_numpy(Zn.reshape(M, K, 13 + m)).to(DEV)
            t = torch.from_numpy(np.tile(np.arange(K) * 0.01, (M, 1))).to(DEV)
            dt = torch.full((M, K), 0.01, dtype=torch.float64, device=DEV)
            fun = RD.DiscretizedDynamics(model, RD.RK4)
            us = timed(lambda i: RD.rollout_(RD.StaticReturn(), fun, {"data": data, "t": t, "dt": dt}), 10, warm=2)
            print(json.dumps({"model": name, "dtype": "f64" if dtype == np.float64 else "f32", "trajectories": M, "knots": K,
                              "us_per_launch": round(us, 2), "us_per_step": round(us / (K - 1), 4), "mode": os.environ.get("RDHIP_TUNE", "rollout=cascade")}), flush=True)
    if "--all" not in sys.argv:
        break
