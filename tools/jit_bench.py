import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "robotdynamics.jl_amd")
import numpy as np, torch, rdhip as RD
from examples.custom_models import SATELLITE_SRC
sys.path.insert(0, "tools")
import bench_configs as B
rng = np.random.default_rng(1)
Zs = B.rigid_inputs(65536, 6, rng)
sat = RD.CustomModel(13, 6, SATELLITE_SRC, params=(1.0, 1.0, 1.0, 1.0), liestate=(RD.QuatRotation, (3, 6)))
B.jac_case("JIT satellite RK4 jacobian ForwardAD (split-role kernel, contiguous plan)", sat, Zs, RD.RK4, RD.ForwardAD(), np.float64, 30)
B.jac_case("JIT satellite RK4 jacobian UserDefined", sat, Zs, RD.RK4, RD.UserDefined(), np.float64, 30)
satb = RD.CustomModel(13, 6, SATELLITE_SRC, params=(1.0, 1.0, 1.0, 1.0), liestate=(RD.QuatRotation, (3, 6))).balance_columns()
B.jac_case("JIT satellite RK4 jacobian ForwardAD (cost-balanced columns)", satb, Zs, RD.RK4, RD.ForwardAD(), np.float64, 30)
B.jac_case("built-in satellite RK4 jacobian ForwardAD", RD.Satellite(), Zs, RD.RK4, RD.ForwardAD(), np.float64, 30)
B.eval_case("JIT satellite RK4 discrete_dynamics", sat, Zs, RD.RK4, np.float64, 30)
B.errexp_case("JIT satellite RK4 error-state expansion (fused, JIT)", sat, Zs, RD.RK4, np.float64, 30)
B.errexp_case("JIT satellite RK4 error-state expansion (fused, JIT, cost-balanced columns)", satb, Zs, RD.RK4, np.float64, 30)
B.errexp_case("built-in satellite RK4 error-state expansion (fused)", RD.Satellite(), Zs, RD.RK4, np.float64, 30)
# a declared block structure (rdhip_model_set_cascade): FiniteDifference along the cascade, rollout with one wavefront per block, block-by-block Newton
CAS = ([2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 0, 0, 0], [2, 2, 2, 0, 0, 0])
satc = RD.CustomModel(13, 6, SATELLITE_SRC, params=(1.0, 1.0, 1.0, 1.0), liestate=(RD.QuatRotation, (3, 6))).balance_columns().set_cascade(*CAS)
ift = RD.ImplicitFunctionTheorem(RD.ForwardAD())
for tag, mdl in (("JIT", satb), ("JIT + cascade", satc), ("built-in", RD.Satellite())):
    B.jac_case(f"{tag} satellite RK4 jacobian FiniteDifference", mdl, Zs, RD.RK4, RD.FiniteDifference(), np.float64, 30)
    B.jac_case(f"{tag} satellite ImplicitMidpoint jacobian (Newton + IFT)", mdl, Zs, RD.ImplicitMidpoint, ift, np.float64, 30)
    B.rollout_case(f"{tag} satellite RK4 rollout fp64 4096 x 101", mdl, 4096, 101, RD.RK4, np.float64, 10, rng)
    B.rollout_case(f"{tag} satellite ImplicitMidpoint rollout fp64 4096 x 101", mdl, 4096, 101, RD.ImplicitMidpoint, np.float64, 5, rng)
