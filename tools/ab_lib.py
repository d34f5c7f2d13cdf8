"""A/B of two builds of librdhip.so on the same box: alternates child processes (RDHIP_LIB) and prints the per-launch times of the
rigid-body Jacobian kernels, outputs rotated over 4 buffers.  usage: python tools/ab_lib.py libA.so libB.so [rounds]"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
import numpy as np, torch
ROOT = os.environ["RD_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import rdhip as RD
import bench_configs as BC
if os.environ.get("RD_SIDE_STREAM"): torch.cuda.set_stream(torch.cuda.Stream(device="cuda:0"))
rng = np.random.default_rng(3)
N = int(os.environ.get("RD_N", "65536"))
sat, quad = RD.Satellite(), RD.Quadrotor()
Zs, Zq = BC.rigid_inputs(N, 6, rng), BC.rigid_inputs(N, 4, rng, 0.6, 1.9)
out = {}
def case(name, model, Zn, rule, diff, reps=400):
    fun = RD.DiscretizedDynamics(model, rule)
    n, m = RD.state_dim(model), RD.control_dim(model)
    Z = RD.PackedTrajectory(n, m, Zn, np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
    Js = [RD.alloc_jacobians(fun, N, Z.data) for _ in range(4)]
    Ys = [torch.empty((N, n), dtype=torch.float64, device="cuda:0") for _ in range(4)]
    calls = [RD.prepare_jacobian(RD.StaticReturn(), diff, fun, Js[b], Ys[b], Z) for b in range(4)]
    BC.timed(lambda i: calls[i % 4](), 7000)          # ~200 ms of load first: clocks settle over ~100 ms, a short warm-up measures the ramp
    out[name] = round(BC.timed(lambda i: calls[i % 4](), reps), 2)
case("sat_rk4_ad", sat, Zs, RD.RK4, RD.ForwardAD())
case("sat_rk4_fd", sat, Zs, RD.RK4, RD.FiniteDifference())
case("sat_rk4_user", sat, Zs, RD.RK4, RD.UserDefined())
case("sat_rk3_ad", sat, Zs, RD.RK3, RD.ForwardAD())
case("quad_rk4_ad", quad, Zq, RD.RK4, RD.ForwardAD())
case("sat_ift", sat, Zs, RD.ImplicitMidpoint, RD.ImplicitFunctionTheorem(RD.ForwardAD()))
print(json.dumps(out))
'''

def main():
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, RDHIP_LIB=os.path.abspath(lib), RD_ROOT=ROOT)
            p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            print(r, os.path.relpath(lib, ROOT), line[-1] if line else p.stderr[-400:], flush=True)

if __name__ == "__main__":
    main()
