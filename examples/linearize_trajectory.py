#!/usr/bin/env python3
"""examples/linearize_trajectory.py -- what a TrajectoryOptimization.jl-style solver does with this library on one GPU:

  1. simulate M candidate control sequences of a quadrotor from the same initial state   (rollout!, src/trajectories.jl:436-441)
  2. linearise every knot point of every trajectory in error-state coordinates           (jacobian! + errstate_jacobian! -> error_expansion!)
  3. do it again every solver iteration on the same arrays                                (HIP graph replay: one launch per iteration)

Run on an MI355X:  python examples/linearize_trajectory.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rdhip as RD  # noqa: E402

M, K, dt = 256, 101, 0.02                      # trajectories, knot points per trajectory, time step
model = RD.Quadrotor()                          # RigidBody{QuatRotation}, n = 13, m = 4 (test/quadrotor.jl)
dmodel = RD.DiscretizedDynamics(model, RD.RK4)
n, m, e = RD.state_dim(model), RD.control_dim(model), RD.errstate_dim(model)
rng = np.random.default_rng(0)

x0, u_hover = RD.zeros(model)                   # identity attitude, zero velocity
u_hover = RD.trim_controls(model)               # thrust that cancels gravity (test/quadrotor.jl:52-54)
data = np.zeros((M, K, n + m))
data[:, :, n:] = u_hover * (1.0 + 0.05 * rng.standard_normal((M, K, m)))     # perturbed hover controls
stack = {"data": torch.as_tensor(data).cuda(),
         "t": torch.as_tensor(np.tile(np.arange(K) * dt, (M, 1))).cuda(),
         "dt": torch.full((M, K), dt, dtype=torch.float64, device="cuda")}
X0 = torch.as_tensor(np.tile(x0, (M, 1))).cuda()

flat = RD.PackedTrajectory(n, m, stack["data"].view(M * K, n + m), stack["t"].view(-1), stack["dt"].view(-1))
E = RD.alloc_error_expansions(dmodel, M * K, flat.data)        # [G2'AG1 | G2'B] per consecutive pair, e x (e+m)


def iteration():
    RD.rollout_(RD.StaticReturn(), dmodel, stack, X0)           # states of knots 2..K of every trajectory
    RD.error_expansion_(dmodel, E, None, flat)                  # fused Jacobian + projection, one kernel


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    iteration()                                                 # first use: creates the context, sets kernel attributes
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        iteration()
    s.synchronize()
    eager = (time.perf_counter() - t0) / 50
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    iteration()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 50

A = E[:, :, :e]                                                 # error-state A_k, B_k views (pairs that straddle two trajectories are unused)
B = E[:, :, e:]
drift = float((stack["data"][:, -1, :3] - X0[:, :3]).norm(dim=1).mean())
print(f"{M} trajectories x {K} knots: {1e6 * eager:.0f} us per iteration eager, {1e6 * graph:.0f} us replayed from a HIP graph")
print(f"mean position drift under perturbed hover after {K * dt:.1f} s: {drift:.3f} m;  |A| max {float(A.abs().max()):.3f}, |B| max {float(B.abs().max()):.3f}")
