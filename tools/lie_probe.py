"""LieSatellite (test/liemodel.jl: n = 7, m = 3) RK4 ForwardAD Jacobian: per-launch time and fraction of the HBM roofline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch, rdhip as RD, bench_configs as BC
rng = np.random.default_rng(2)
model = RD.LieSatellite()
n, m = RD.state_dim(model), RD.control_dim(model)
for N in (65536, 1 << 20):
    q = rng.standard_normal((N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    Zn = np.concatenate([rng.random((N, 3)), q, rng.random((N, 3))], axis=1)
    fun = RD.DiscretizedDynamics(model, RD.RK4)
    Z = RD.PackedTrajectory(n, m, Zn, np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
    nb = 4 if N == 65536 else 2
    Js = [RD.alloc_jacobians(fun, N, Z.data) for _ in range(nb)]
    Ys = [torch.empty((N, n), dtype=torch.float64, device="cuda:0") for _ in range(nb)]
    calls = [RD.prepare_jacobian(RD.StaticReturn(), RD.ForwardAD(), fun, Js[b], Ys[b], Z) for b in range(nb)]
    us = BC.timed(lambda i: calls[i % nb](), 200)
    by = N * ((n + m) * 8 + 16 + n * (n + m) * 8 + n * 8)
    print(f"LieSatellite RK4 ForwardAD N={N}: {us:.2f} us  frac {by / (us * 1e-6) / 8e12:.3f}", flush=True)
