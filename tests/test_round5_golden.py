"""CPU side of the round-5 fixtures (tests/golden/make_golden_round5.py; ref_*.npz from make_golden_reference.jl replace the expected values key by key
when a Julia install has produced them): the oracle against the definition-derived values the GPU tests hold the kernels to."""
import numpy as np

from oracle import rd_oracle as O
from tests import helpers as H

TOL = 1e-10


def test_oracle_user_defined_cartpole_jacobian_is_the_jacobian():
    """UserDefined() on the cartpole: the hand-written formulas of test/cartpole_model.jl:57-96 (oracle/rd_oracle.cpp user_jacobian) and the
    integrators' chain rule on them (src/integration.jl:149-177, 302-337) against forward-mode derivatives of the definitions."""
    d = H.load("userjac_cartpole")
    M = O.Model(O.CARTPOLE, [float(v) for v in d["params"]])
    Z, t, dt = d["Z"], d["t"], d["dt"]
    for rule, key in ((O.CONTINUOUS, "Jc_user"), (O.EULER, "Jd_user_euler"), (O.RK2, "Jd_user_rk2"), (O.RK3, "Jd_user_rk3"), (O.RK4, "Jd_user_rk4")):
        J, _ = M.jacobian(rule, O.USER, Z, t, dt)
        np.testing.assert_allclose(J, d[key], rtol=0, atol=TOL * max(1.0, np.abs(d[key]).max()))


def test_discrete_double_integrator_fixture_is_self_consistent():
    """test/discrete_dynamics_test.jl:35-73: x+ = [p + v h; v + u h], J = [I h I 0; 0 I h I] in the reference's column order, J2 = [-I 0]."""
    d = H.load("discrete_di")
    Z, dt = d["Z"], d["dt"]
    for k in range(Z.shape[0]):
        A = np.eye(4, 6); A[0, 2] = A[1, 3] = A[2, 4] = A[3, 5] = dt[k]
        np.testing.assert_allclose(A @ Z[k], d["xn"][k], rtol=0, atol=1e-15)
        np.testing.assert_allclose(d["J_user"][k], A, rtol=0, atol=1e-15)
        np.testing.assert_allclose(d["J_ad"][k], A, rtol=0, atol=1e-15)
    np.testing.assert_allclose(d["e"], d["xn"][:-1] - Z[1:, :4], rtol=0, atol=1e-15)
    assert (d["J2"][:, range(4), range(4)] == -1).all() and np.abs(d["J2"]).sum() == 4 * (Z.shape[0] - 1)


def test_oracle_propagates_a_nan_knot_point_and_leaves_the_others():
    d = H.load("nan_satellite")
    M = O.Model(O.SATELLITE, np.concatenate([[float(d["mass"])], d["J"].flatten()]))
    J, Y = M.jacobian(O.RK4, O.FORWARD_AD, d["Z"], d["t"], d["dt"])
    bad = np.concatenate([d["poisoned"], d["inf_dt"]])
    good = np.setdiff1d(np.arange(d["Z"].shape[0]), bad)
    np.testing.assert_allclose(J[good], d["Jd_rk4"][good], rtol=0, atol=TOL)
    np.testing.assert_allclose(Y[good], d["xn_rk4"][good], rtol=0, atol=TOL)
    assert np.isnan(Y[int(d["poisoned"][0])]).all()
