"""rollout timings of the rigid bodies (explicit RK4 / RK3 and ImplicitMidpoint), HIP-event timed."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import numpy as np, rdhip as RD, bench_configs as BC
rng = np.random.default_rng(5)
out = {}
def emit(name, N, dt, us, alg, extra=None): out[name] = round(us, 1)
BC.emit = emit
sat = RD.Satellite()
BC.rollout_case("sat 4096x101 f64", sat, 4096, 101, RD.RK4, np.float64, 10, rng)
BC.rollout_case("sat 512x256 f32", sat, 512, 256, RD.RK4, np.float32, 10, rng)
BC.rollout_case("sat body 4096x101 f64", RD.Satellite(bodyframe=True), 4096, 101, RD.RK4, np.float64, 10, rng)
BC.rollout_case("quad 4096x101 f64", RD.Quadrotor(), 4096, 101, RD.RK4, np.float64, 10, rng)
BC.rollout_case("sat rk3 4096x101 f64", sat, 4096, 101, RD.RK3, np.float64, 10, rng)
BC.rollout_case("sat 16384x64 f64", sat, 16384, 64, RD.RK4, np.float64, 10, rng)
BC.rollout_case("sat 65536x32 f64", sat, 65536, 32, RD.RK4, np.float64, 5, rng)
BC.rollout_case("sat implicit 4096x101 f64", sat, 4096, 101, RD.ImplicitMidpoint, np.float64, 5, rng)
BC.rollout_case("sat implicit 512x256 f32", sat, 512, 256, RD.ImplicitMidpoint, np.float32, 5, rng)
print(json.dumps(out))
