import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robotdynamics.jl_amd"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")]
import numpy as np, torch, rdhip as RD
import bench as B
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
Mr, Kr = 512, 256
def build(satgen):
    funs, stacks, Jc = [], [], []
    for mdl, mctl in ((RD.Cartpole(), 1), (RD.Satellite(), 6)):
        n = RD.state_dim(mdl)
        zz = satgen(Mr * Kr) if n == 13 else np.random.default_rng(14).random((Mr * Kr, n + mctl))
        fun = RD.DiscretizedDynamics(mdl, RD.RK4)
        st = {"data": torch.as_tensor(zz.astype(np.float32).reshape(Kr, Mr, n + mctl)).to(dev),
              "t": torch.as_tensor(np.ascontiguousarray(np.broadcast_to((np.arange(Kr) * 0.01)[:, None], (Kr, Mr)))).to(dev),
              "dt": torch.full((Kr, Mr), 0.01, dtype=torch.float64, device=dev)}
        funs.append(fun); stacks.append(st); Jc.append(RD.alloc_jacobians(fun, Mr * Kr, st["data"]))
    return RD.PreparedRolloutJacobianMulti(RD.StaticReturn(), RD.ForwardAD(), funs, Jc, stacks, None, time_major=True), stacks
import bench_configs as BC
rng = np.random.default_rng(3)
for name, gen in (("bench synthetic_inputs", lambda n: B.synthetic_inputs(n, 13)[0]), ("configs rigid_inputs", lambda n: BC.rigid_inputs(n, 6, rng))):
    c5, stacks = build(gen)
    for reps in (20, 100):
        us = B._timed_us(lambda i: c5(), reps, warm=2)
        print(name, "reps", reps, "us", round(us, 1), "finite", bool(torch.isfinite(stacks[1]["data"]).all()), bool(torch.isfinite(stacks[0]["data"]).all()), flush=True)
