"""The kernels that are NOT the flagship, a few launches each, for the HBM byte counters (tools/pmc_kernels.sh runs this under
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate passes).  Every case prints one JSON line with the kernel names it launches and the
ALGORITHMIC bytes per launch (SURVEY 8d's per-unit figures x the units of the launch), which the shell script sets against the counters.
    python tools/pmc_kernels.py            # all cases
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import numpy as np
import torch

import rdhip as RD
import bench_configs as BC

REPS = 4
rng = np.random.default_rng(11)
sig, AD = RD.StaticReturn(), RD.ForwardAD()
sat, cart, quad = RD.Satellite(), RD.Cartpole(), RD.Quadrotor()
dsat, dcart = RD.DiscretizedDynamics(sat, RD.RK4), RD.DiscretizedDynamics(cart, RD.RK4)


def report(case, kernels, note=""):
    """kernels: {substring of the kernel name: algorithmic bytes per launch of that kernel}"""
    print(json.dumps({"case": case, "reps": REPS, "kernels": kernels, "note": note}), flush=True)


def tm_stack(model, M, K, dtype):
    n, m = RD.state_dim(model), RD.control_dim(model)
    data = (0.3 * BC.rigid_inputs(M * K, m, rng) + np.array([0] * 3 + [0.7, 0, 0, 0] + [0] * (6 + m)) if n == 13 else 0.2 * rng.random((M * K, n + m))).reshape(K, M, n + m).astype(dtype)
    t = np.repeat(np.arange(K)[:, None] * 0.01, M, axis=1); dt = np.full((K, M), 0.01)
    return {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(t).cuda(), "dt": torch.as_tensor(dt).cuda()}


# 1. BASELINE config 5 on one GPU in one call: 4 096 + 4 096 trajectories x 256 knots, fp32, time-major
M, K = 4096, 256
funs = [dsat, dcart]
stacks = [tm_stack(sat, M, K, np.float32), tm_stack(cart, M, K, np.float32)]
Js = [RD.alloc_jacobians(f, K * M, s["data"]) for f, s in zip(funs, stacks)]
call = RD.PreparedRolloutJacobianMulti(sig, AD, funs, Js, stacks, None, time_major=True)
for _ in range(REPS):
    call()
torch.cuda.synchronize()
NK = M * K
report("c5_whole_one_call_fp32", {
    # round 6: ONE launch (mixed.hpp).  satellite: the rollout reads u, t, dt (24 + 16) and writes x (52), the Jacobian role reads z, t, dt (76 + 16) and writes
    # J (988); cartpole: the rollout reads u, t, dt (4 + 16) and writes x (16), the Jacobians read z, t, dt (20 + 16) and write J (80)
    "rollout_jac_mixed_kernel": NK * (24 + 16 + 52 + 76 + 16 + 988) + NK * (4 + 16 + 16 + 20 + 16 + 80)}, "both models in one launch")
RD.tune("mixed=0")                                       # round 5's form of the same call: the satellite's fused launch + the cartpole in four pipelined pieces
for _ in range(REPS):
    call()
torch.cuda.synchronize()
RD.tune("reset")
report("c5_whole_streams_fp32", {
    "rollout_jac_fused_kernel": NK * (24 + 16 + 52 + 76 + 16 + 988),
    "rollout_kernel": NK * (4 + 16 + 16) // 4,
    "jac_ad_small_kernel": NK * (20 + 16 + 80) // 4}, "RDHIP_TUNE mixed=0; cartpole kernels: per piece (a quarter of the stack)")
del stacks, Js, call
torch.cuda.empty_cache()

# 2. errstate_jacobian! at BASELINE config 4's size, both modes
N = 262144
X = torch.as_tensor(BC.rigid_inputs(N, 4, rng)[:, :13]).cuda()
G = RD.alloc_errstate_jacobians(quad, N, X)
for _ in range(REPS):
    RD.errstate_jacobian_(quad, G, X, clear=True)
torch.cuda.synchronize()
for _ in range(REPS):
    RD.errstate_jacobian_(quad, G, X)
torch.cuda.synchronize()
report("errstate_jacobian_262144", {"errstate_tile_kernel": N * (104 + 1248), "errstate_pattern_kernel": N * (104 + 21 * 8)},
       "pattern kernel (clear = 0): 21 scattered values per 1 248-byte slab, every line read-modify-written: traffic above the algorithmic bytes is inherent")
del X, G

# 3. ImplicitMidpoint + IFT, 4. fp32 Jacobians at 1 M knots, 6. many short trajectories in one launch
N = 65536
Z = RD.PackedTrajectory(13, 6, BC.rigid_inputs(N, 6, rng), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda")
dmi = RD.DiscretizedDynamics(sat, RD.ImplicitMidpoint)
J = RD.alloc_jacobians(dmi, N, Z.data); Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
for _ in range(REPS):
    RD.jacobian_(sig, RD.ImplicitFunctionTheorem(AD), dmi, J, Y, Z)
torch.cuda.synchronize()
report("implicit_midpoint_ift_65536", {"implicit_ift_split_kernel": N * 2248})
del Z, J, Y
N = 1 << 20
Z = RD.PackedTrajectory(13, 6, BC.rigid_inputs(N, 6, rng).astype(np.float32), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda")
J = RD.alloc_jacobians(dsat, N, Z.data)
for _ in range(REPS):
    RD.jacobian_(sig, AD, dsat, J, None, Z)
torch.cuda.synchronize()
report("jacobian_fp32_1M", {"jac_ad_split_kernel": N * 1080})
del Z, J
torch.cuda.empty_cache()
nb, Nb = 1024, 101
Zs = [RD.PackedTrajectory(13, 6, BC.rigid_inputs(Nb, 6, rng), np.arange(Nb) * 0.1, np.full(Nb, 0.1)).to("cuda") for _ in range(nb)]
Jm = [RD.alloc_jacobians(dsat, Nb, Zs[0].data) for _ in range(nb)]
Ym = [torch.empty((Nb, 13), dtype=torch.float64, device="cuda") for _ in range(nb)]
for _ in range(REPS):
    RD.jacobian_multi_(sig, AD, dsat, Jm, Ym, Zs)
torch.cuda.synchronize()
report("jacobian_multi_1024x101", {"jac_split_multi_kernel": nb * Nb * 2248})
del Zs, Jm, Ym

# 5. the streaming rollout of big batches (rollout_stream.hpp): 32 768 trajectories x 32 knots, fp64, time-major -- x written, u / t / dt read per step;
#    the kernel re-writes the controls of the lines it touches (rdhip.h: whole-line stores)
M, K = 32768, 32
st = tm_stack(sat, M, K, np.float64)
for _ in range(REPS):
    RD.rollout_(sig, dsat, st, None, time_major=True)
torch.cuda.synchronize()
report("rollout_stream_32768x32", {"rollout_stream_kernel": M * (K - 1) * (48 + 16 + 104) + M * 104},
       "algorithmic: per step read u, t, dt (64 B) and write x (104 B); the kernel moves whole knot points (152 B lines in and out)")

# 5b. the same rollout on split arrays (rollout_split.hpp): U, t, dt read once, X written once
X = torch.empty((K, M, 13), dtype=torch.float64, device="cuda"); X[0] = st["data"][0, :, :13]
U = st["data"][:K - 1, :, 13:].contiguous()
for _ in range(REPS):
    RD.rollout_split_(sig, dsat, X, U, st["t"], st["dt"], time_major=True)
torch.cuda.synchronize()
report("rollout_split_32768x32", {"rollout_split_lane_kernel": M * (K - 1) * (48 + 16 + 104) + M * 104}, "time-major")
Xj = torch.empty((M, K, 13), dtype=torch.float64, device="cuda"); Xj[:, 0] = st["data"][0, :, :13]
Uj = st["data"][:K - 1, :, 13:].transpose(0, 1).contiguous()
tj, dj = st["t"].t().contiguous(), st["dt"].t().contiguous()
for _ in range(REPS + 1):                                  # (one more launch than the time-major case: the script tells the two apart by their launch counts)
    RD.rollout_split_(sig, dsat, Xj, Uj, tj, dj)
torch.cuda.synchronize()
report("rollout_split_32768x32_trajectory_major", {"rollout_split_lane_kernel": M * (K - 1) * (48 + 16 + 104) + M * 104}, "trajectory-major: a trajectory's 4-knot runs start and end inside 128-byte lines")
