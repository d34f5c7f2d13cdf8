"""CPU: pin the oracle against every known-answer fact the reference's own tests hold for the hot path
(SURVEY.md section 8c).  File:line citations are into /root/reference."""
import math

import numpy as np
import pytest

from oracle import rd_oracle as O
from tests import helpers as H


# ---- integer facts (bit-exact) --------------------------------------------------------------
def test_liestate_index_tables():
    # test/liestate.jl:33-48 : QuatRotation, P = (3,2,3)
    ln, ed, rf, vf = O.liestate_query(O.ROT_QUAT, (3, 2, 3))
    assert (ln, ed) == (16, 14)
    assert rf == [4, 10] and vf == [1, 8, 14]
    # test/liestate.jl:56-67 : RodriguesParam, P = (3,2,3)
    ln, ed, rf, vf = O.liestate_query(O.ROT_RP, (3, 2, 3))
    assert (ln, ed) == (14, 14)
    assert rf == [4, 9] and vf == [1, 7, 12]
    # test/liemodel.jl:24-26 : QuatState(7,(4,)) == LieState(Quat,(3,0)), n = 7, e = 6
    assert O.liestate_query(O.ROT_QUAT, (3, 0))[:2] == (7, 6)
    # examples/satellite_pair.jl:49,58 : (0,3,3) -> 14 (quat), 12 (MRP)
    assert O.liestate_query(O.ROT_QUAT, (0, 3, 3))[0] == 14
    assert O.liestate_query(O.ROT_MRP, (0, 3, 3))[0] == 12


def test_rigidbody_dims_and_inds():
    M = O.body()
    assert M.dims() == (13, 6, 13) and M.e == 12            # test/rigidbody_test.jl:60, rigid_body_jacobians.jl:5,73
    assert M.liestate() == (O.ROT_QUAT, (3, 6))              # test/rigidbody_test.jl:69
    g = M.gen_inds()                                         # test/rigidbody_test.jl:88-92
    assert list(g["q"]) == [4, 5, 6, 7] and list(g["u"]) == list(range(14, 20)) and list(g["r"]) == [1, 2, 3]
    assert list(g["v"]) == [8, 9, 10] and list(g["w"]) == [11, 12, 13]
    M2 = O.Model(O.SATELLITE, M.params, rot=O.ROT_MRP)       # test/rigidbody_test.jl:94-98
    g = M2.gen_inds()
    assert list(g["q"]) == [4, 5, 6] and list(g["v"]) == [7, 8, 9] and list(g["u"]) == list(range(13, 19))
    for rot, rp in ((O.ROT_QUAT, 4), (O.ROT_MRP, 3), (O.ROT_RP, 3)):   # test/rigidbody_test.jl:79-85
        assert O.Model(O.SATELLITE, M.params, rot=rot).n == 9 + rp
    assert O.quadrotor().dims() == (13, 4, 13)
    assert O.cartpole().dims() == (4, 1, 4) and O.cartpole().e == 4
    assert O.satellite_pair().dims() == (14, 6, 14) and O.satellite_pair().e == 12
    assert O.lie_satellite().dims() == (7, 3, 7) and O.lie_satellite().e == 6


# ---- closed-form Jacobians -----------------------------------------------------------------
@pytest.mark.parametrize("D", [1, 2, 3])
def test_double_integrator_closed_forms(D):
    M = O.double_integrator(D)
    rng = np.random.default_rng(D)
    Z = rng.random((5, 3 * D))
    for diff in (O.FORWARD_AD, O.USER, O.FINITE_DIFF):
        J, y = M.jacobian(O.CONTINUOUS, diff, Z, 0.0, 0.1)
        J0 = np.zeros((2 * D, 3 * D))
        for i in range(2 * D):
            J0[i, i + D] = 1.0                                # test/double_integrator.jl:119-127
        np.testing.assert_allclose(J, np.broadcast_to(J0, J.shape), atol=1e-7 if diff == O.FINITE_DIFF else 0)
        np.testing.assert_array_equal(y, np.concatenate([Z[:, D:2 * D], Z[:, 2 * D:]], axis=1))
    # Euler-discretised DI == test/discrete_dynamics_test.jl:35-63 : x+ = [p + v dt; v + u dt]
    dt = 0.05
    xn = M.discrete_dynamics(O.EULER, Z, 0.0, dt)
    np.testing.assert_allclose(xn, np.concatenate([Z[:, :D] + Z[:, D:2 * D] * dt, Z[:, D:2 * D] + Z[:, 2 * D:] * dt], axis=1), atol=1e-15)
    J, _ = M.jacobian(O.EULER, O.FORWARD_AD, Z, 0.0, dt)
    J0 = np.zeros((2 * D, 3 * D))
    for i in range(2 * D):
        J0[i, i] = 1.0
        J0[i, i + D] = dt
    np.testing.assert_allclose(J, np.broadcast_to(J0, J.shape), atol=1e-15)


def test_linear_system_rk4_is_truncated_exponential():
    # for xdot = Ax + Bu, RK4 gives A_d = sum_{k<=4} (hA)^k/k!, B_d = h (I + hA/2 + (hA)^2/6 + (hA)^3/24) B
    rng = np.random.default_rng(7)
    n, m, h = 6, 3, 0.1
    A, B = rng.standard_normal((n, n)), rng.standard_normal((n, m))
    M = O.linear(A, B)
    Z = rng.random((4, n + m))
    hA = h * A
    Ad = np.eye(n) + hA + hA @ hA / 2 + hA @ hA @ hA / 6 + hA @ hA @ hA @ hA / 24
    Bd = h * (np.eye(n) + hA / 2 + hA @ hA / 6 + hA @ hA @ hA / 24) @ B
    for diff in (O.FORWARD_AD, O.USER):
        J, xn = M.jacobian(O.RK4, diff, Z, 0.0, h)
        np.testing.assert_allclose(J[:, :, :n], np.broadcast_to(Ad, (4, n, n)), atol=1e-13)
        np.testing.assert_allclose(J[:, :, n:], np.broadcast_to(Bd, (4, n, m)), atol=1e-13)
        np.testing.assert_allclose(xn, Z[:, :n] @ Ad.T + Z[:, n:] @ Bd.T, atol=1e-13)
    # RK3: up to 3rd order, RK2: 2nd order, Euler: 1st order
    for rule, K in ((O.EULER, 1), (O.RK2, 2), (O.RK3, 3)):
        Ad = sum(np.linalg.matrix_power(hA, k) / math.factorial(k) for k in range(K + 1))
        Bd = h * sum(np.linalg.matrix_power(hA, k) / math.factorial(k + 1) for k in range(K)) @ B
        J, _ = M.jacobian(rule, O.FORWARD_AD, Z, 0.0, h)
        np.testing.assert_allclose(J[0], np.concatenate([Ad, Bd], axis=1), atol=1e-13)


def test_rk_step_formulas_cartpole():
    # test/integration_tests.jl:38 (Euler), :49-59 (RK4), :81-99 (RK3), :42-45 (RK2, commented)
    M = O.cartpole()
    rng = np.random.default_rng(3)
    z = rng.random((1, 5))
    x, dt = z[0, :4], 0.01
    f = lambda xx: M.dynamics(np.concatenate([xx, z[0, 4:]])[None], 0.0, dt)[0]
    np.testing.assert_allclose(M.discrete_dynamics(O.EULER, z, 0.0, dt)[0], x + f(x) * dt, rtol=1e-14)
    k1 = f(x) * dt; k2 = f(x + k1 / 2) * dt
    np.testing.assert_allclose(M.discrete_dynamics(O.RK2, z, 0.0, dt)[0], x + k2, rtol=1e-14)
    k3 = f(x - k1 + 2 * k2) * dt
    np.testing.assert_allclose(M.discrete_dynamics(O.RK3, z, 0.0, dt)[0], x + (k1 + 4 * k2 + k3) / 6, rtol=1e-14)
    k3 = f(x + k2 / 2) * dt; k4 = f(x + k3) * dt
    np.testing.assert_allclose(M.discrete_dynamics(O.RK4, z, 0.0, dt)[0], x + (k1 + 2 * k2 + 2 * k3 + k4) / 6, rtol=1e-14)


def test_all_six_sig_diff_combinations_agree():
    # check_jacobians of test/integration_tests.jl:7-18
    M = O.cartpole()
    z = np.random.default_rng(5).random((3, 5))
    for rule in (O.EULER, O.RK3, O.RK4):
        J0, _ = M.jacobian(rule, O.FORWARD_AD, z, 0.0, 0.01)
        for sig in (O.STATIC_RETURN, O.INPLACE):
            for diff in (O.FORWARD_AD, O.FINITE_DIFF, O.USER):
                J, _ = M.jacobian(rule, diff, z, 0.0, 0.01, sig)
                np.testing.assert_allclose(J, J0, rtol=0, atol=1e-6 if diff == O.FINITE_DIFF else 1e-10)


# ---- rigid body ---------------------------------------------------------------------------
def _qrot(q, r):
    w, v = q[0], q[1:]
    return (w * w - v @ v) * r + 2 * v * (v @ r) + 2 * w * np.cross(v, r)


def test_newton_euler_components():
    # test/rigidbody_test.jl:135-162 with the Body model (mass 2, J = diag(2,3,1))
    rng = np.random.default_rng(11)
    Z = H.rand_rigid(rng, 6, 6)
    mass, J = 2.0, np.diag([2.0, 3.0, 1.0])
    for frame in (O.FRAME_WORLD, O.FRAME_BODY):
        xd = O.body(frame).dynamics(Z, 0.0, 0.01)
        for k in range(6):
            r, q, v, w, u = Z[k, :3], Z[k, 3:7], Z[k, 7:10], Z[k, 10:13], Z[k, 13:]
            F, T = _qrot(q, u[:3]), u[3:]
            L = np.array([[q[0], -q[1], -q[2], -q[3]], [q[1], q[0], -q[3], q[2]], [q[2], q[3], q[0], -q[1]], [q[3], -q[2], q[1], q[0]]])
            np.testing.assert_allclose(xd[k, 3:7], 0.5 * L @ np.concatenate([[0], w]), atol=1e-15)   # kinematics
            np.testing.assert_allclose(xd[k, 10:13], np.linalg.solve(J, T - np.cross(w, J @ w)), atol=1e-14)
            if frame == O.FRAME_WORLD:
                np.testing.assert_allclose(xd[k, :3], v, atol=0)
                np.testing.assert_allclose(xd[k, 7:10], F / mass, atol=1e-15)
            else:
                qc = q * np.array([1, -1, -1, -1])
                np.testing.assert_allclose(xd[k, :3], _qrot(q, v), atol=1e-15)
                np.testing.assert_allclose(xd[k, 7:10], _qrot(qc, F / mass) - np.cross(w, v), atol=1e-15)


def test_errstate_jacobian_block_structure():
    # test/rigid_body_jacobians.jl:72-83 : G = cat(I3, L(q)H, I6), nablaG = cat(0_3, -(q.b) I3, 0_6)
    rng = np.random.default_rng(13)
    Z = H.rand_rigid(rng, 5, 6)
    M = O.body()
    G = M.errstate_jacobian(Z[:, :13])
    b = rng.random((5, 13))
    dG = M.nabla_errstate_jacobian(Z[:, :13], b)
    assert G.shape == (5, 13, 12) and dG.shape == (5, 12, 12)
    for k in range(5):
        q = Z[k, 3:7]
        L = np.array([[q[0], -q[1], -q[2], -q[3]], [q[1], q[0], -q[3], q[2]], [q[2], q[3], q[0], -q[1]], [q[3], -q[2], q[1], q[0]]])
        G0 = np.zeros((13, 12)); G0[:3, :3] = np.eye(3); G0[3:7, 3:6] = L[:, 1:]; G0[7:, 6:] = np.eye(6)
        np.testing.assert_allclose(G[k], G0, atol=1e-15)
        D0 = np.zeros((12, 12)); D0[3:6, 3:6] = -(q @ b[k, 3:7]) * np.eye(3)
        np.testing.assert_allclose(dG[k], D0, atol=1e-15)
    # the reference does NOT clear G (src/liestate.jl:293-297): pre-existing entries outside the pattern survive
    G1 = M.errstate_jacobian(Z[:, :13], clear=False, G0=np.full((5, 13, 12), 7.0))
    assert G1[0, 0, 5] == 7.0 and G1[0, 0, 0] == 1.0
    # Euclidean model: G = I with clearing (src/statevectortype.jl:149-155), nablaG = 0, state_diff = x - x0
    C = O.cartpole()
    np.testing.assert_array_equal(C.errstate_jacobian(rng.random((2, 4))), np.broadcast_to(np.eye(4), (2, 4, 4)))
    x, x0 = rng.random((2, 4)), rng.random((2, 4))
    np.testing.assert_array_equal(C.state_diff(x, x0), x - x0)


def test_state_diff_matches_rodrigues_of_relative_rotation():
    # test/rbstate.jl:157 ; test/rigid_body_jacobians.jl:64-70
    rng = np.random.default_rng(17)
    X, X0 = H.rand_rigid(rng, 4, 6)[:, :13], H.rand_rigid(rng, 4, 6)[:, :13]
    d = O.body().state_diff(X, X0)
    np.testing.assert_allclose(d[:, :3], (X - X0)[:, :3], atol=0)
    np.testing.assert_allclose(d[:, 6:], (X - X0)[:, 7:], atol=0)
    for k in range(4):
        q, q0 = X[k, 3:7], X0[k, 3:7]
        L0 = np.array([[q0[0], q0[1], q0[2], q0[3]], [-q0[1], q0[0], q0[3], -q0[2]], [-q0[2], -q0[3], q0[0], q0[1]], [-q0[3], q0[2], -q0[1], q0[0]]])
        dq = L0 @ q
        np.testing.assert_allclose(d[k, 3:6], dq[1:] / dq[0], atol=1e-14)


def test_terminal_knot_point_reads_zero_control():
    # src/knotpoint.jl:67,123 : dt === 0 -> control(z) = zeros ; test/knotpoints.jl:29-31
    M = O.cartpole()
    z = np.array([[0.1, 0.2, 0.3, 0.4, 5.0]])
    z0 = z.copy(); z0[0, 4] = 0.0
    np.testing.assert_array_equal(M.dynamics(z, 0.0, 0.0), M.dynamics(z0, 0.0, 0.1))
    J, y = M.jacobian(O.RK4, O.FORWARD_AD, z, 0.0, 0.0)
    np.testing.assert_array_equal(J[0], np.concatenate([np.eye(4), np.zeros((4, 1))], axis=1))
    np.testing.assert_array_equal(y[0], z[0, :4])


def test_rollout_matches_repeated_steps():
    # src/trajectories.jl:436-441 ; test/trajectories.jl:184-189 (StaticReturn ~ InPlace)
    M = O.cartpole()
    rng = np.random.default_rng(19)
    Z = rng.random((3, 11, 5))
    dt = np.full((3, 11), 0.1); dt[:, -1] = 0.0
    t = np.arange(11) * 0.1
    R = M.rollout(O.RK4, Z, t, dt)
    x = Z[:, 0, :4].copy()
    for k in range(1, 11):
        x = M.discrete_dynamics(O.RK4, np.concatenate([x, Z[:, k - 1, 4:]], axis=1), t[k - 1], 0.1)
        np.testing.assert_allclose(R[:, k, :4], x, atol=1e-14)
    np.testing.assert_array_equal(R[:, :, 4:], Z[:, :, 4:])
    np.testing.assert_allclose(M.rollout(O.RK4, Z, t, dt, sig=O.INPLACE), R, atol=1e-12)


def test_implicit_midpoint_linear_closed_form_and_properties():
    """ImplicitMidpoint (src/integration.jl:620-729).  Known answers independent of any implementation:
    * linear system xdot = A x + B u:  x2 = (I - hA/2)^-1 ((I + hA/2) x1 + h B u),  d x2/d x1 = (I - hA/2)^-1 (I + hA/2),  d x2/d u = h (I - hA/2)^-1 B,
      Newton converges in one step;
    * any model: the solution satisfies x1 + h f((x1+x2)/2, u) - x2 = 0 (test/implicit_dynamics_test.jl:46-60) and the IFT Jacobian agrees with
      central differences of the solve."""
    rng = np.random.default_rng(17)
    A, B = rng.standard_normal((4, 4)), rng.standard_normal((4, 2))
    m = O.linear(A, B)
    Z = rng.random((6, 6)); h = 0.1
    Xn, J, its = m.implicit_midpoint(Z, 0.0, h)
    I = np.eye(4)
    Minv = np.linalg.inv(I - h * A / 2)
    np.testing.assert_allclose(Xn, (Minv @ ((I + h * A / 2) @ Z[:, :4].T + h * B @ Z[:, 4:].T)).T, rtol=0, atol=1e-13)
    np.testing.assert_allclose(J[:, :, :4], np.broadcast_to(Minv @ (I + h * A / 2), (6, 4, 4)), rtol=0, atol=1e-13)
    np.testing.assert_allclose(J[:, :, 4:], np.broadcast_to(h * Minv @ B, (6, 4, 2)), rtol=0, atol=1e-13)
    assert its.max() <= 1
    for om, Zn in ((O.cartpole(), rng.random((5, 5))), (O.satellite(), H.rand_rigid(rng, 5, 6))):
        n = om.n
        Xn, J, its = om.implicit_midpoint(Zn, 0.0, h)
        xm = np.concatenate([(Zn[:, :n] + Xn) / 2, Zn[:, n:]], axis=1)
        assert np.abs(Zn[:, :n] + h * om.dynamics(xm, 0.0, h) - Xn).max() < 1e-13
        eps = 1e-6
        for j in range(Zn.shape[1]):
            Zp, Zm = Zn.copy(), Zn.copy(); Zp[:, j] += eps; Zm[:, j] -= eps
            fd = (om.implicit_midpoint(Zp, 0.0, h)[0] - om.implicit_midpoint(Zm, 0.0, h)[0]) / (2 * eps)
            np.testing.assert_allclose(J[:, :, j], fd, rtol=0, atol=2e-8)
        E, J2, J1 = om.implicit_midpoint_error(np.concatenate([Zn[:1], np.concatenate([Xn[:1], Zn[:1, n:]], axis=1)]), 0.0, h)
        assert np.abs(E).max() < 1e-13                       # dynamics_error(z2 = [x_next; u], z1) = 0
        np.testing.assert_allclose(-np.linalg.solve(J2[0][:, :n], J1[0]), J[0], rtol=0, atol=1e-12)    # J = -J2[:,1:n] \\ J1


def test_wrenches_and_their_jacobian_closed_forms():
    """wrenches(model, x, u, t) and wrench_jacobian! (src/rigidbody.jl:244-248, :315-320).  Known answers:
    * test/rigidbody_test.jl:26-46 (`Body`): F = q * u[1:3], M = u[4:6]; dF/dq = grad-rotate(q, f) (here: central differences of the rotation of
      the NON-normalised quaternion, as `QuatRotation(q, false) * v` in test/rigidbody_test.jl:21), dF/du[1:3] = RotMatrix(q), dM/du[4:6] = I, zero elsewhere;
    * test/rigidbody_test.jl:143-148: linear_velocity(xdot) = F / mass and angular_velocity(xdot) = inv(J) (T - w x J w) for every RigidBody;
    * the quadrotor's wrench (test/quadrotor.jl:56-96): gravity + thrust along the body z axis, moments from the arm length / torque constant."""
    rng = np.random.default_rng(23)
    N = 7
    Z = H.rand_rigid(rng, N, 6)

    def rotm(q):
        w, x, y, z = q
        return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])      # q v q* without normalising q
    om = O.body()
    W, F = om.wrenches(Z, 0.0), om.wrench_jacobian(Z, 0.0)
    assert W.shape == (N, 6) and F.shape == (N, 6, 19)
    for k in range(N):
        q, f = Z[k, 3:7], Z[k, 13:16]
        np.testing.assert_allclose(W[k, :3], rotm(q) @ f, rtol=0, atol=1e-14)
        np.testing.assert_array_equal(W[k, 3:], Z[k, 16:19])
        np.testing.assert_allclose(F[k, :3, 13:16], rotm(q), rtol=0, atol=1e-14)
        np.testing.assert_array_equal(F[k, 3:, 16:19], np.eye(3))
        for j in range(4):
            e = np.zeros(4); e[j] = 1e-6
            np.testing.assert_allclose(F[k, :3, 3 + j], (rotm(q + e) @ f - rotm(q - e) @ f) / 2e-6, rtol=0, atol=1e-8)
        mask = np.ones((6, 19), bool); mask[:3, 3:7] = False; mask[:3, 13:16] = False; mask[3:, 16:19] = False
        assert np.all(F[k][mask] == 0.0)
    # Newton-Euler identities for every rigid body (both frames enter only through the dynamics, not the wrench)
    for om, Zr, mass, J in ((O.body(), Z, 2.0, np.diag([2.0, 3.0, 1.0])), (O.quadrotor(), H.rand_rigid(rng, N, 4, 0.5, 1.5), 0.5, np.diag([0.0023, 0.0023, 0.004]))):
        W = om.wrenches(Zr, 0.0)
        xd = om.dynamics(Zr, 0.0, 0.1)
        w = Zr[:, 10:13]
        np.testing.assert_allclose(xd[:, 7:10], W[:, :3] / mass, rtol=0, atol=1e-12)
        np.testing.assert_allclose(xd[:, 10:13], np.linalg.solve(J, (W[:, 3:] - np.cross(w, w @ J.T)).T).T, rtol=0, atol=1e-10)
        Fj = om.wrench_jacobian(Zr, 0.0)
        for j in range(Zr.shape[1]):                          # central differences of the wrench itself
            Zp, Zm = Zr.copy(), Zr.copy(); Zp[:, j] += 1e-6; Zm[:, j] -= 1e-6
            np.testing.assert_allclose(Fj[:, :, j], (om.wrenches(Zp, 0.0) - om.wrenches(Zm, 0.0)) / 2e-6, rtol=0, atol=1e-7)
    # three-parameter attitudes act through their unit quaternion
    for rot in (O.ROT_MRP, O.ROT_RP):
        om3 = O.satellite(2.0, np.diag([2.0, 3.0, 1.0]), O.FRAME_WORLD, rot)
        Z3 = np.concatenate([rng.random((N, 3)), 0.6 * rng.random((N, 3)) - 0.3, rng.random((N, 6)), rng.random((N, 6))], axis=1)
        W3 = om3.wrenches(Z3, 0.0)
        for k in range(N):
            p = Z3[k, 3:6]; n2 = p @ p
            q = np.concatenate([[1 - n2], 2 * p]) / (1 + n2) if rot == O.ROT_MRP else np.concatenate([[1.0], p]) / np.sqrt(1 + n2)
            np.testing.assert_allclose(W3[k, :3], rotm(q) @ Z3[k, 12:15], rtol=0, atol=1e-13)
