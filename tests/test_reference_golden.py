"""Parity pin.  `tests/golden/ref_*.npz` are produced by the real RobotDynamics.jl (tests/golden/make_golden_reference.jl, one command on any
machine with Julia).  The build image has no Julia, so normally none exist and this file reports the parity status as what it is --
"restatement-derived" -- instead of silently passing; once they exist, the oracle (and, through tests/helpers.load, every golden test of the
HIP path) is held to the reference's own numbers at the north-star tolerance."""
import os

import numpy as np
import pytest

from oracle import rd_oracle as O
from tests import helpers as H

TOL = 1e-10
REFS = H.reference_fixtures()


def test_generator_is_committed_and_covers_every_fixture():
    src = open(os.path.join(H.GOLDEN, "make_golden_reference.jl")).read()
    for needle in ("using RobotDynamics", "npzwrite", "dyn_case", "exp_case", "err_case", "FiniteDifference", "MRP{Float64}", "RodriguesParam{Float64}",
                   "RD.errstate_jacobian!", "RD.∇errstate_jacobian!", "RD.state_diff", "RD.dynamics_error", "DiscretizedDynamics{Q}",
                   "RD.ImplicitMidpoint", "ImplicitFunctionTheorem", "RD.rollout!", "TVPendulum", "TVSat", "RobotDynamicsHIP", "host_smoke",
                   "DiscreteDI <: RobotDynamics.DiscreteDynamics", "discrete_case()", "user_jacobian_case()", "nan_case()", "RD.dynamics_error_jacobian!"):
        assert needle in src, needle
    for key in ("xdot", "Jc", "xn_$rn", "Jd_$rn", "Jfd_rk4", "E_$rn", "err_$rn", '"G"', '"dG"', '"dX"', "xn_implicit", "J_ift", "ref_rollout_", "ref_tv_",
                "ref_discrete_di.npz", "ref_userjac_cartpole.npz", "ref_nan_satellite.npz", "Jd_user_$rn", "J_user"):
        assert key in src, key


@pytest.mark.skipif(bool(REFS), reason="reference-generated fixtures present")
def test_parity_status_is_restatement_derived():
    pytest.skip("parity unpinned against the Julia reference: no tests/golden/ref_*.npz (run tests/golden/make_golden_reference.jl where Julia is installed)")


@pytest.mark.parametrize("name", [r[4:] for r in REFS if r.startswith("dyn_")])
def test_oracle_matches_the_reference_dynamics(name):
    d = H.load("dyn_" + name)
    assert bool(d["_reference_derived"])
    M = H.oracle_model(name, d)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    np.testing.assert_allclose(M.dynamics(Z, t, dt), d["xdot"], rtol=0, atol=TOL)
    for rn, rule in (("euler", O.EULER), ("rk3", O.RK3), ("rk4", O.RK4)):
        J, Y = M.jacobian(rule, O.FORWARD_AD, Z, t, dt)
        np.testing.assert_allclose(Y, d["xn_" + rn], rtol=0, atol=TOL)
        np.testing.assert_allclose(J, d["Jd_" + rn], rtol=0, atol=TOL * max(1.0, np.abs(d["Jd_" + rn]).max()))
    if "W" in d:            # RigidBody models: wrenches(model, x, u, t) and its ForwardDiff Jacobian
        np.testing.assert_allclose(M.wrenches(Z, t), d["W"], rtol=0, atol=TOL)
        np.testing.assert_allclose(M.wrench_jacobian(Z, t), d["Fw"], rtol=0, atol=TOL)
    if "Jfd_rk4" in d:      # FiniteDiff's step rule: the reference's own tests only hold 1e-6; report how close the restated rule is
        J, _ = M.jacobian(O.RK4, O.FINITE_DIFF, Z, t, dt)
        np.testing.assert_allclose(J, d["Jfd_rk4"], rtol=0, atol=1e-6 * max(1.0, np.abs(d["Jd_rk4"]).max()))


# (err_* and exp_* reference fixtures need no test of their own: tests/test_oracle_golden.py and the GPU golden tests load them through
# tests/helpers.load, which substitutes the reference's values key by key)


@pytest.mark.parametrize("tag", [r[:-5] for r in REFS if r in ("mrp_body", "rp_body")])
def test_three_parameter_rotations_state_diff(tag):
    """RigidBody{MRP} / RigidBody{RodriguesParam}: the product serves their dynamics, explicit / implicit rules, Jacobians, rollouts,
    errstate_jacobian! and state_diff on the [3P] formulas of SURVEY App. A, which so far are pinned by their definitions only
    (tests/golden/make_golden_rot3.py); with reference values present every one of them is held to the reference here."""
    d = dict(np.load(os.path.join(H.GOLDEN, f"ref_{tag}_body.npz")))
    rot = O.ROT_MRP if tag == "mrp" else O.ROT_RP
    M = O.Model(O.SATELLITE, np.concatenate([[2.0], np.diag([2.0, 3.0, 1.0]).flatten()]), rot=rot)
    Z, t, dt = d["Z"], d["t"], d["dt"]
    X = Z[:, :12]
    np.testing.assert_allclose(M.state_diff(X, np.roll(X, -1, axis=0)), d["dX"], rtol=0, atol=TOL)
    np.testing.assert_allclose(M.dynamics(Z, t, dt), d["xdot"], rtol=0, atol=TOL)
    J, Y = M.jacobian(O.RK4, O.FORWARD_AD, Z, t, dt)
    np.testing.assert_allclose(Y, d["xn_rk4"], rtol=0, atol=TOL)
    np.testing.assert_allclose(J, d["Jd_rk4"], rtol=0, atol=TOL * max(1.0, np.abs(d["Jd_rk4"]).max()))
    np.testing.assert_allclose(M.errstate_jacobian(X), d["G"], rtol=0, atol=TOL)


@pytest.mark.parametrize("name", [r[9:] for r in REFS if r.startswith("implicit_")])
def test_oracle_matches_the_reference_implicit_midpoint(name):
    d = dict(np.load(os.path.join(H.GOLDEN, f"ref_implicit_{name}.npz")))
    M = H.oracle_model(name, H.load("dyn_" + name))
    Xn, J, _ = M.implicit_midpoint(d["Z"], d["t"], d["dt"])
    np.testing.assert_allclose(Xn, d["xn_implicit"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(J, d["J_ift"], rtol=0, atol=1e-8 * max(1.0, np.abs(d["J_ift"]).max()))


@pytest.mark.parametrize("name", [r[8:] for r in REFS if r.startswith("rollout_") and not r.endswith("_implicit")])
def test_oracle_matches_the_reference_rollout(name):
    d = dict(np.load(os.path.join(H.GOLDEN, f"ref_rollout_{name}.npz")))
    fixture, rn = name.rsplit("_", 1)
    M = H.oracle_model(fixture, H.load("dyn_" + fixture))
    K = d["Z"].shape[0]
    X = M.rollout({"rk4": O.RK4, "rk3": O.RK3}[rn], d["Z"][None], d["t"][None], d["dt"][None], X0=d["Z"][:1, :d["X"].shape[1]])[0]
    np.testing.assert_allclose(X[:, :d["X"].shape[1]], d["X"], rtol=0, atol=TOL)
    assert K == d["X"].shape[0]


@pytest.mark.parametrize("tag", [r[3:] for r in REFS if r.startswith("tv_")])
def test_oracle_matches_the_reference_time_dependent_models(tag):
    d = dict(np.load(os.path.join(H.GOLDEN, f"ref_tv_{tag}.npz")))
    M = O.Model(O.TV_PENDULUM, [2.0, 0.7, 3.0, 0.1]) if tag == "pendulum" else O.Model(O.TV_SATELLITE, np.concatenate([[2.0], np.diag([2.0, 3.0, 1.0]).flatten(), [0.3, 0.2, 1.7]]))
    Z, t, dt = d["Z"], d["t"], d["dt"]
    np.testing.assert_allclose(M.dynamics(Z, t, dt), d["xdot"], rtol=0, atol=TOL)
    J, Y = M.jacobian(O.RK4, O.FORWARD_AD, Z, t, dt)
    np.testing.assert_allclose(Y, d["xn_rk4"], rtol=0, atol=TOL)
    np.testing.assert_allclose(J, d["Jd_rk4"], rtol=0, atol=TOL * max(1.0, np.abs(d["Jd_rk4"]).max()))
    np.testing.assert_allclose(M.jacobian(O.RK3, O.FORWARD_AD, Z, t, dt)[1], d["xn_rk3"], rtol=0, atol=TOL)
