"""The flagship at 1 M / 2 M / 4 M knot points, four launches each, in this order (tools/pmc_bign.sh reads the counters of the launches by order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch, rdhip as RD, bench_configs as BC
rng = np.random.default_rng(5)
sat = RD.Satellite()
dm = RD.DiscretizedDynamics(sat, RD.RK4)
for N in (1 << 20, 1 << 21, 1 << 22):
    Zn = BC.rigid_inputs(N, 6, rng)
    Z = RD.PackedTrajectory(13, 6, Zn, np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
    Js = [RD.alloc_jacobians(dm, N, Z.data) for _ in range(2)]
    Ys = [torch.empty((N, 13), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    for i in range(4):
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, Js[i % 2], Ys[i % 2], Z)
    torch.cuda.synchronize()
    del Z, Js, Ys
    torch.cuda.empty_cache()
