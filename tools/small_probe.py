"""Single small batches (one trajectory of 101 .. 8 192 knot points): the column-split kernel (default below 4 tiles per CU) against the tile loop
(RDHIP_TUNE=small=tile), satellite / quadrotor RK4 ForwardAD Jacobian + x_next, bare C calls, outputs rotated."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import rdhip as RD
import bench_configs as BC
rng = np.random.default_rng(2)
tag = os.environ.get("RDHIP_TUNE", "colsplit")
def emit(name, N, dt, us, alg, extra=None): print(json.dumps({"mode": tag, "name": name, "N": N, "us": round(us, 2), "frac": round(alg / us / 1e6 / 8, 3)}), flush=True)
BC.emit = emit
sat, quad = RD.Satellite(), RD.Quadrotor()
for N in (101, 512, 1024, 2048, 4096, 8192, 12288, 16384, 32768):
    BC.jac_case("satellite RK4 AD", sat, BC.rigid_inputs(N, 6, rng), RD.RK4, RD.ForwardAD(), np.float64, 300)
BC.jac_case("quadrotor RK4 AD", quad, BC.rigid_inputs(4096, 4, rng, 0.6, 1.9), RD.RK4, RD.ForwardAD(), np.float64, 300)
BC.jac_case("satellite RK4 AD fp32", sat, BC.rigid_inputs(4096, 6, rng), RD.RK4, RD.ForwardAD(), np.float32, 300)
BC.jac_case("satellite RK3 AD", sat, BC.rigid_inputs(4096, 6, rng), RD.RK3, RD.ForwardAD(), np.float64, 300)
