"""ImplicitMidpoint IFT Jacobian: one output buffer (stays in the 256 MiB Infinity Cache) vs outputs rotated over 4 buffers (545 MB)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")]
import rdhip as RD
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as BC
DEV = torch.device("cuda:0")
rng = np.random.default_rng(3)
N = 65536
Zs = BC.rigid_inputs(N, 6, rng)
sat = RD.Satellite()
dmi = RD.DiscretizedDynamics(sat, RD.ImplicitMidpoint)
dm4 = RD.DiscretizedDynamics(sat, RD.RK4)
Z = RD.PackedTrajectory(13, 6, Zs, np.arange(N) * 0.1, np.full(N, 0.1)).to(DEV)
ift = RD.ImplicitFunctionTheorem(RD.ForwardAD())
BUFS = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4]
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for nb in BUFS:
    Js = [RD.alloc_jacobians(dmi, N, Z.data) for _ in range(nb)]
    Ys = [torch.empty((N, 13), dtype=torch.float64, device=DEV) for _ in range(nb)]
    for name, fun, diff in (("ift", dmi, ift), ("rk4_ad", dm4, RD.ForwardAD()), ("rk4_fd", dm4, RD.FiniteDifference())):
        calls = BC.bare([lambda b=b: RD.jacobian_(RD.StaticReturn(), diff, fun, Js[b], Ys[b], Z) for b in range(nb)])
        us = BC.timed(lambda i: calls[i % nb](), REPS)
        prep = [RD.prepare_jacobian(RD.StaticReturn(), diff, fun, Js[b], Ys[b], Z) for b in range(nb)]
        us2 = BC.timed(lambda i: prep[i % nb](), REPS)
        print(json.dumps({"case": name, "buffers": nb, "bare_us": us, "prepared_us": us2}), flush=True)
