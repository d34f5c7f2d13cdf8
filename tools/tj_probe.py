import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import rdhip as RD
import bench_configs as BC
M, K = 4096, 256
rng = np.random.default_rng(7)
model, mctl, n = RD.Satellite(), 6, 13
Zn = BC.rigid_inputs(M * K, mctl, rng).astype(np.float32)
fun = RD.DiscretizedDynamics(model, RD.RK4)
tk = np.arange(K) * 0.01
for tm in (True, False):
    shp = (K, M) if tm else (M, K)
    st = {"data": torch.from_numpy(Zn.reshape(shp + (n + mctl,))).to("cuda:0"),
          "t": torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[:, None] if tm else tk[None, :], shp))).to("cuda:0"),
          "dt": torch.full(shp, 0.01, dtype=torch.float64, device="cuda:0")}
    J = RD.alloc_jacobians(fun, M * K, st["data"])
    for knob in ("", "fused_tj_max_knots=100000000"):
        RD.tune(knob if knob else "reset")
        l = RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), fun, J, st, None, time_major=tm)
        us = BC.timed(lambda i: RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), fun, J, st, None, time_major=tm), 20, warm=2)
        print(f"satellite 4096x256 fp32 time_major={tm} knob={knob!r}: launches {l} {us:.1f} us  {RD.last_kernel_name()[:60]}")
        RD.tune("reset")
