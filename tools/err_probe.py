"""errstate_jacobian! (dense G): knot points per tile (RDHIP_ERR_TK tuning switch), C4's per-GPU share (32 768) and the 262 144-knot batch."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import rdhip as RD
from rdhip import _lib as L
import bench_configs as BC
lib = L.lib()
dev = torch.device("cuda:0")
quad = RD.Quadrotor()
for Nq in (32768, 262144):
    rng = np.random.default_rng(4)
    q = rng.standard_normal((Nq, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = torch.as_tensor(np.concatenate([rng.random((Nq, 3)), q, rng.random((Nq, 6))], axis=1)).to(dev)
    nb = 16 if Nq < 100000 else 2
    Gs = [RD.alloc_errstate_jacobians(quad, Nq, X) for _ in range(nb)]
    ctx = L.context_for(0, torch.cuda.current_stream(dev).cuda_stream)
    f = lambda i: lib.rdhip_errstate_jacobian(ctx.handle, quad._handle, L.F64, Nq, X.data_ptr(), 13, 1, Gs[i % nb].data_ptr(), L.DEVICE)
    BC.timed(f, 3000 if Nq > 100000 else 20000)
    for r in range(2):
        us = BC.timed(f, 300)
        print(os.environ.get("RDHIP_ERR_TK", "64"), Nq, round(us, 2), "us", round(Nq * 1352 / us / 1e6 / 8, 3))
