#!/usr/bin/env python3
"""tools/multi_vs_single.py -- does a ONE-batch rdhip_jacobian_multi call cost what rdhip_jacobian costs?  (Round 6: can the single-batch split-role
kernels be dropped in favour of the batch-table kernels?)  One JSON line per (body, dtype, N, entry point)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rdhip as RD  # noqa: E402
from bench_configs import rigid_inputs, timed, nbuf_for, DEV  # noqa: E402


def case(model, name, diff, dtype, N, rule=RD.RK4, reps=200):
    rng = np.random.default_rng(3)
    n, m = RD.state_dim(model), RD.control_dim(model)
    T = 8 if dtype == np.float64 else 4
    Zn = rigid_inputs(N, m, rng) if n == 13 else rng.random((N, n + m))
    fun = RD.DiscretizedDynamics(model, rule)
    nb = nbuf_for(N * n * (n + m) * T)
    Zs = [RD.PackedTrajectory(n, m, Zn.astype(dtype), np.arange(N) * 0.1, np.full(N, 0.1)).to(DEV) for _ in range(nb)]
    Js = [RD.alloc_jacobians(fun, N, Zs[0].data) for _ in range(nb)]
    Ys = [torch.empty((N, n), dtype=Zs[0].data.dtype, device=DEV) for _ in range(nb)]
    sig = RD.StaticReturn()
    single = [RD.prepare_jacobian(sig, diff, fun, Js[b], Ys[b], Zs[b]) for b in range(nb)]
    multi = [RD.prepare_jacobian_multi(sig, diff, fun, [Js[b]], [Ys[b]], [Zs[b]]) for b in range(nb)]
    alg = N * ((n + m) * T + 16 + n * (n + m) * T + n * T)
    out = {}
    for rnd in range(2):
        for tag, calls in (("single", single), ("multi1", multi)):
            us = timed(lambda i: calls[i % nb](), reps)
            out.setdefault(tag, []).append(round(us, 2))
    ref = Js[0].clone()
    single[0](); torch.cuda.synchronize(); a = Js[0].clone()
    Js[0].zero_(); multi[0](); torch.cuda.synchronize()
    same = bool(torch.equal(a, Js[0]))
    print(json.dumps({"name": name, "dtype": "f64" if T == 8 else "f32", "N": N, "us": out, "frac_single": alg / (min(out["single"]) * 1e-6) / 8e12,
                      "frac_multi1": alg / (min(out["multi1"]) * 1e-6) / 8e12, "bit_identical": same}), flush=True)
    del ref


if __name__ == "__main__":
    sat = RD.Satellite()
    quadb = RD.Quadrotor(bodyframe=True)
    for N in (65536, 262144):
        case(sat, "satellite RK4 ForwardAD", RD.ForwardAD(), np.float64, N)
        case(sat, "satellite RK4 FiniteDifference", RD.FiniteDifference(), np.float64, N)
        case(sat, "satellite RK4 UserDefined", RD.UserDefined(), np.float64, N)
        case(sat, "satellite RK4 ForwardAD", RD.ForwardAD(), np.float32, N)
        case(quadb, "quadrotor body RK4 ForwardAD", RD.ForwardAD(), np.float64, N)
    case(sat, "satellite RK4 ForwardAD", RD.ForwardAD(), np.float64, 1 << 20, reps=40)
