"""The flagship at 1 M / 2 M / 4 M knot points, repeated in one process (fresh allocations per repetition)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch, rdhip as RD, bench_configs as BC
rng = np.random.default_rng(5)
res = []
def emit(name, N, dt, us, alg, extra=None): res.append((N, round(us, 1), round(alg / us / 1e6 / 8, 3)))
BC.emit = emit
sat = RD.Satellite()
for rep in range(3):
    for N in (1 << 20, 1 << 21, 1 << 22):
        BC.jac_case("x", sat, BC.rigid_inputs(N, 6, rng), RD.RK4, RD.ForwardAD(), np.float64, 8)
        torch.cuda.empty_cache()
print(os.environ.get("RD_TAG", ""), res)
