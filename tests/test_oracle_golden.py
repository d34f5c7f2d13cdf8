"""CPU: the C++ oracle against the committed torch-autograd golden vectors (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import rd_oracle as O
from tests import helpers as H

TOL = 1e-10   # fp64 parity tolerance of BASELINE.json's north_star


@pytest.mark.parametrize("name", H.dyn_cases())
def test_dynamics_and_steps(name):
    d = H.load("dyn_" + name)
    M = H.oracle_model(name, d)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    np.testing.assert_allclose(M.dynamics(Z, t, dt), d["xdot"], rtol=0, atol=TOL)
    for rn, rule in H.RULES.items():
        for sig in (O.STATIC_RETURN, O.INPLACE):
            np.testing.assert_allclose(M.discrete_dynamics(rule, Z, t, dt, sig), d["xn_" + rn], rtol=0, atol=TOL)


@pytest.mark.parametrize("name", H.dyn_cases())
def test_forward_ad_jacobians(name):
    d = H.load("dyn_" + name)
    M = H.oracle_model(name, d)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    J, Y = M.jacobian(O.CONTINUOUS, O.FORWARD_AD, Z, t, dt)
    np.testing.assert_allclose(J, d["Jc"], rtol=0, atol=TOL)
    np.testing.assert_allclose(Y, d["xdot"], rtol=0, atol=TOL)
    for rn, rule in H.RULES.items():
        J, Y = M.jacobian(rule, O.FORWARD_AD, Z, t, dt)
        np.testing.assert_allclose(J, d["Jd_" + rn], rtol=0, atol=TOL)
        np.testing.assert_allclose(Y, d["xn_" + rn], rtol=0, atol=TOL)


@pytest.mark.parametrize("name", H.dyn_cases())
def test_finite_difference_within_1e6(name):
    # reference tolerance for FiniteDifference: atol = 1e-6  (test/integration_tests.jl:14)
    d = H.load("dyn_" + name)
    M = H.oracle_model(name, d)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    J, _ = M.jacobian(O.CONTINUOUS, O.FINITE_DIFF, Z, t, dt)
    scale = max(1.0, np.abs(d["Jc"]).max())
    np.testing.assert_allclose(J, d["Jc"], rtol=0, atol=1e-6 * scale)
    J, _ = M.jacobian(O.RK4, O.FINITE_DIFF, Z, t, dt)
    np.testing.assert_allclose(J, d["Jd_rk4"], rtol=0, atol=1e-6 * scale)


@pytest.mark.parametrize("name", [n for n in H.dyn_cases() if not n.startswith(("satpair", "liesat")) and not H.three_parameter(n)])
def test_user_defined_chain_rule(name):
    # chain rule == AD through the step to 1e-10 (test/integration_tests.jl:7-18,64-71,102-109)
    d = H.load("dyn_" + name)
    M = H.oracle_model(name, d)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    J, _ = M.jacobian(O.CONTINUOUS, O.USER, Z, t, dt)
    np.testing.assert_allclose(J, d["Jc"], rtol=0, atol=TOL * max(1.0, np.abs(d["Jc"]).max()))
    for rn, rule in H.RULES.items():
        J, Y = M.jacobian(rule, O.USER, Z, t, dt)
        np.testing.assert_allclose(J, d["Jd_" + rn], rtol=0, atol=TOL * max(1.0, np.abs(d["Jd_" + rn]).max()))
        np.testing.assert_allclose(Y, d["xn_" + rn], rtol=0, atol=TOL)


def test_user_defined_missing_raises():
    M = O.satellite_pair()
    with pytest.raises(NotImplementedError):
        M.jacobian(O.CONTINUOUS, O.USER, np.ones((1, 20)), 0.0, 0.1)
    with pytest.raises(NotImplementedError):                                      # the analytic rigid-body Jacobian is for QuatRotation only (src/rigidbody.jl:260)
        O.satellite(rot=O.ROT_MRP).jacobian(O.RK4, O.USER, np.ones((1, 18)), 0.0, 0.1)


@pytest.mark.parametrize("rtag,rot", [("mrp", O.ROT_MRP), ("rp", O.ROT_RP)])
def test_errstate_three_parameter_rotations(rtag, rot):
    """grad-differential of MRP / RodriguesParam (oracle closed forms) against the blocks differentiated from the composition map
    params(R * R(delta)) in tests/golden/make_golden_rot3.py; Cayley state difference through scipy."""
    d = H.load(f"err_rigid_{rtag}")
    M = O.satellite(2.0, np.diag([2.0, 3.0, 1.0]), rot=rot)
    assert (M.n, M.e) == (12, 12)                                                  # test/rigidbody_test.jl:94-98
    np.testing.assert_allclose(M.errstate_jacobian(d["X"]), d["G"], rtol=0, atol=TOL)
    np.testing.assert_allclose(M.state_diff(d["X"], d["X0"]), d["dX"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("tag,mk", [("rigid", O.satellite), ("satpair", O.satellite_pair), ("liesat", O.lie_satellite)])
def test_errstate(tag, mk):
    d = H.load("err_" + tag)
    M = mk()
    np.testing.assert_allclose(M.errstate_jacobian(d["X"]), d["G"], rtol=0, atol=TOL)
    np.testing.assert_allclose(M.nabla_errstate_jacobian(d["X"], d["Xbar"]), d["dG"], rtol=0, atol=TOL)
    np.testing.assert_allclose(M.state_diff(d["X"], d["X0"]), d["dX"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("rtag,rot", [("mrp", O.ROT_MRP), ("rp", O.ROT_RP)])
@pytest.mark.parametrize("tag", ["rigid", "p323"])
def test_nabla_errstate_three_parameter_rotations(tag, rtag, rot):
    """grad^2-differential of MRP / RodriguesParam (src/liestate.jl:300-320 calls it per rotation block; the oracle's closed forms) against the Hessian of
    delta -> xbar . (x (+) delta) differentiated twice from the composition map (tests/golden/make_golden_rot3.py second_stage) -- the definition under
    which the quaternion fixtures reproduce Rotations.jl's -(q . b) I."""
    d = H.load(f"nab_{tag}_{rtag}")
    P = tuple(int(v) for v in d["P"])
    M = O.satellite(2.0, np.diag([2.0, 3.0, 1.0]), rot=rot) if P == (3, 6) else None
    if M is None:
        pytest.skip("the oracle's registry has no LieState{R,(3,2,3)} model; the GPU test covers it through a custom LieState model")
    H_ = M.nabla_errstate_jacobian(d["X"], d["Xbar"])
    np.testing.assert_allclose(H_, d["dG"], rtol=0, atol=TOL * max(1.0, np.abs(d["dG"]).max()))
    assert np.abs(H_ - H_.transpose(0, 2, 1)).max() == 0 or np.allclose(H_, H_.transpose(0, 2, 1), atol=1e-14)     # a Hessian contraction is symmetric


EXP_CASES = ["cartpole", "satellite_world", "body_body", "satfull_world", "quadrotor_world", "satpair", "liesat"] + \
            [f"{k}_{r}_{f}" for k in ("satellite", "satfull", "quadrotor") for r in ("mrp", "rp") for f in ("world", "body")]


@pytest.mark.parametrize("name", EXP_CASES)
def test_dynamics_error_and_error_expansion_against_the_definition(name):
    """exp_*.npz: E_k = G(x_{k+1})' d f_d(x_k (+) delta, u)/d(delta, u) and e_k = f_d(z_k) - x_{k+1}, differentiated by torch from the
    composition map itself (tests/golden/make_golden.py) -- independent of the oracle's G2' [A G1 | B] product."""
    d, g = H.load("dyn_" + name), H.load("exp_" + name)
    M = H.oracle_model(name, d)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    for rn, rule in (("rk3", O.RK3), ("rk4", O.RK4)):
        np.testing.assert_allclose(M.dynamics_error(rule, Z, t, dt), g["err_" + rn], rtol=0, atol=TOL)
        E, _ = M.error_expansion(rule, Z, t, dt)
        np.testing.assert_allclose(E, g["E_" + rn], rtol=0, atol=TOL * max(1.0, np.abs(g["E_" + rn]).max()))
        J2, J1, e = M.dynamics_error_jacobian(rule, O.FORWARD_AD, Z, t, dt)
        np.testing.assert_allclose(J1, d["Jd_" + rn][:-1], rtol=0, atol=TOL * max(1.0, np.abs(d["Jd_" + rn]).max()))
        np.testing.assert_allclose(e, g["err_" + rn], rtol=0, atol=TOL)
