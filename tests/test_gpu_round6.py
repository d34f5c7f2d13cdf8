"""GPU tests added in round 6: state_diff with the non-default error maps (src/liestate.jl:26-32,216-217), the split-layout rollout, and the hardened
fused-launch ordering."""
import os

import numpy as np
import pytest

import rdhip as RD
from oracle import rd_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-10
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MAPS = {"cayley": RD.CayleyMap, "exponential": RD.ExponentialMap, "mrp": RD.MRPMap, "quatvec": RD.QuatVecMap, "identity": RD.IdentityMap}
ROT = {0: RD.QuatRotation, 1: RD.MRP, 2: RD.RodriguesParam}


def host(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)


# ------------------------------------------------------------------------------------------------------------------------------------------
# state_diff(model, x, x0, errmap): three LieStates x five maps against the geometry-derived fixtures and the oracle, fp64 at 1e-10
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("where", ["host", "cuda"])
@pytest.mark.parametrize("case", ["errmap_rigid", "errmap_p323", "errmap_p323_mrp", "errmap_rigid_rp"])
def test_state_diff_error_maps_against_fixtures_and_oracle(case, where):
    d = H.load(case)
    s = RD.LieState(ROT[int(d["rot"])], tuple(int(v) for v in d["P"]))
    om = H.errmap_oracle_model(d)
    X, X0 = d["X"], d["X0"]
    if where == "cuda":
        X, X0 = torch.as_tensor(X).cuda(), torch.as_tensor(X0).cuda()
    for name, Map in MAPS.items():
        if "dX_" + name not in d:                     # IdentityMap of a QuatRotation state: Rotations.rotation_error throws (four values into a RotationError)
            with pytest.raises(NotImplementedError):
                RD.state_diff(s, X, X0, Map())
            continue
        got = host(RD.state_diff(s, X, X0, Map()))
        np.testing.assert_allclose(got, d["dX_" + name], rtol=TOL, atol=TOL, err_msg=f"{case} {name} vs fixture")
        np.testing.assert_allclose(got, om.state_diff(d["X"], d["X0"], errmap=Map.code), rtol=TOL, atol=TOL, err_msg=f"{case} {name} vs oracle")
    # the default is CayleyMap, and the class itself is accepted like an instance (Julia passes a singleton)
    assert np.array_equal(host(RD.state_diff(s, X, X0)), host(RD.state_diff(s, X, X0, RD.CayleyMap)))
    with pytest.raises(TypeError):
        RD.state_diff(s, X, X0, "cayley")


@pytest.mark.parametrize("rot", [0, 1, 2])
def test_state_diff_error_maps_model_states_large_batch_and_fp32(rot):
    """the RigidBody state (3, 6) of the three rotation types through a MODEL (state_diff(model, x, x0, errmap), src/liestate.jl:26-28), 70 001 states read
    out of packed trajectories (leading dimension n + m), every map, fp64 vs the oracle at 1e-10 and fp32 vs the fp64 oracle at rel 1e-4"""
    rng = np.random.default_rng(660 + rot)
    model = RD.Quadrotor(R=ROT[rot])
    om = O.quadrotor(rot=rot)
    n, m = RD.state_dim(model), RD.control_dim(model)
    N = 70001
    Z, Z0 = rng.random((N, n + m)) - 0.5, rng.random((N, n + m)) - 0.5
    if rot == 0:
        Z[:, 3:7], Z0[:, 3:7] = rng.standard_normal((N, 4)), rng.standard_normal((N, 4))
    for name, Map in MAPS.items():
        if rot == 0 and name == "identity":
            continue
        exp = om.state_diff(Z, Z0, errmap=Map.code)
        tr = lambda A, dt: RD.PackedTrajectory(n, m, torch.as_tensor(A.astype(dt)).cuda(), torch.zeros(N, dtype=torch.float64, device="cuda"), torch.ones(N, dtype=torch.float64, device="cuda"))
        got = host(RD.state_diff(model, tr(Z, np.float64), tr(Z0, np.float64), Map()))
        np.testing.assert_allclose(got, exp, rtol=TOL, atol=TOL, err_msg=f"rot {rot} {name} fp64")
        exp32 = om.state_diff(Z.astype(np.float32).astype(np.float64), Z0.astype(np.float32).astype(np.float64), errmap=Map.code)
        got32 = host(RD.state_diff(model, tr(Z, np.float32), tr(Z0, np.float32), Map())).astype(np.float64)
        # (relative to the size of the rotation error of the knot point: v / w and the shadow-set branches are unbounded near 180 degrees)
        scale = np.maximum(1.0, np.abs(exp32).max(axis=1, keepdims=True))
        assert np.max(np.abs(got32 - exp32) / scale) < 2e-4, f"rot {rot} {name} fp32"


def test_state_diff_error_maps_wide_leading_dimension_takes_the_elementwise_kernel():
    """leading dimensions above 32 (states inside a wider record) bypass the tiled kernel: the element-wise one, every map"""
    rng = np.random.default_rng(67)
    s = RD.LieState(RD.QuatRotation, (3, 2, 3))
    om = O.lie_state(0, (3, 2, 3))
    n, N, ld = len(s), 1000, 40
    W, W0 = rng.standard_normal((N, ld)), rng.standard_normal((N, ld))
    Xd, X0d = torch.as_tensor(W).cuda()[:, :n], torch.as_tensor(W0).cuda()[:, :n]
    for name, Map in MAPS.items():
        if name == "identity":
            continue
        got = host(RD.state_diff(s, Xd, X0d, Map()))
        assert "state_diff_kernel" in RD.last_kernel_name()
        np.testing.assert_allclose(got, om.state_diff(W[:, :n], W0[:, :n], errmap=Map.code), rtol=TOL, atol=TOL, err_msg=name)
