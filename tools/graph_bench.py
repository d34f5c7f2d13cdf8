#!/usr/bin/env python3
"""tools/graph_bench.py -- launch-bound regime: one solver-style iteration (RK4 Jacobians + error-state Jacobians + fused error-state
expansion of ONE short trajectory) issued eagerly vs replayed from a HIP graph (torch.cuda.graph captures the library's launches on
the capture stream; the library does nothing but enqueue kernels when the data is device resident)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rdhip as RD  # noqa: E402


def run(N, iters_per_graph=10, reps=200):
    rng = np.random.default_rng(0)
    model = RD.Satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    q = rng.standard_normal((N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    Zn = np.concatenate([rng.random((N, 3)), q, rng.random((N, 12))], axis=1)
    Z = RD.PackedTrajectory(13, 6, Zn, np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
    J = RD.alloc_jacobians(dm, N, Z.data)
    Y = torch.empty((N, 13), dtype=torch.float64, device="cuda:0")
    G = RD.alloc_errstate_jacobians(model, N, Z.data)
    E = RD.alloc_error_expansions(dm, N, Z.data)
    X = Z.data[:, :13]
    sig, diff = RD.StaticReturn(), RD.ForwardAD()

    def iteration():
        RD.jacobian_(sig, diff, dm, J, Y, Z)
        RD.errstate_jacobian_(model, G, X, clear=True)
        RD.error_expansion_(dm, E, None, Z)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(20):
            iteration()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps * iters_per_graph):
            iteration()
        s.synchronize()
        eager = (time.perf_counter() - t0) / (reps * iters_per_graph) * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(iters_per_graph):
            iteration()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (reps * iters_per_graph) * 1e6
    print(json.dumps({"name": "satellite RK4: jacobian! + errstate_jacobian! + error expansion (3 kernels per iteration)", "N": N,
                      "eager_us_per_iteration": round(eager, 2), "hip_graph_us_per_iteration": round(graph, 2),
                      "iterations_per_graph": iters_per_graph}))


if __name__ == "__main__":
    for N in (101, 1024, 4096):
        run(N)
