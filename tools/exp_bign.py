#!/usr/bin/env python3
"""tools/exp_bign.py -- developer experiment: why does the flagship kernel drop from 0.75 to 0.62 of the roofline between 1 M and 2 M knot points?
Times the Jacobian launch and a plain torch fill of the same output buffer at each size (same event method)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import rdhip as RD
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import rigid_inputs, timed

torch.cuda.set_device(0)
rng = np.random.default_rng(7)
sat = RD.Satellite(); fun = RD.DiscretizedDynamics(sat, RD.RK4)
for p in (18, 19, 20, 21, 22):
    N = 1 << p
    Z = RD.PackedTrajectory(13, 6, rigid_inputs(N, 6, rng), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
    for nb in (1, 2):
        Js = [RD.alloc_jacobians(fun, N, Z.data) for _ in range(nb)]
        Ys = [torch.empty((N, 13), dtype=torch.float64, device="cuda:0") for _ in range(nb)]
        calls = [RD.prepare_jacobian(RD.StaticReturn(), RD.ForwardAD(), fun, Js[b], Ys[b], Z) for b in range(nb)]
        reps = max(8, (1 << 24) // N)
        us = timed(lambda i: calls[i % nb](), reps)
        base = [J.transpose(1, 2) if not J.is_contiguous() else J for J in Js]
        flat = [torch.as_strided(J, (J.numel(),), (1,)) for J in Js]
        usf = timed(lambda i: flat[i % nb].fill_(1.0), reps)
        alg = N * 2248
        print(json.dumps({"N": N, "nbuf": nb, "reps": reps, "jac_us": round(us, 2), "jac_frac": alg / (us * 1e-6) / 8e12,
                          "fill_us": round(usf, 2), "fill_TBps": N * 1976 / (usf * 1e-6) / 1e12}), flush=True)
        del Js, Ys, calls, flat, base
        torch.cuda.empty_cache()
