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
        # (v / w and the shadow-set branches are unbounded near 180 degrees and lose digits like 1 / w^2 there: fp32 is held to rel 2e-4 where the rotation
        # error stays below 30, and to finiteness beyond)
        big = np.abs(exp32).max(axis=1)
        ok = big < 30.0
        scale = np.maximum(1.0, big[ok, None])
        assert ok.mean() > 0.9 and np.isfinite(got32).all()
        assert np.max(np.abs(got32[ok] - exp32[ok]) / scale) < 2e-4, f"rot {rot} {name} fp32"


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


# ------------------------------------------------------------------------------------------------------------------------------------------
# rdhip_rollout_split: states and controls in arrays of their own (src/trajectories.jl:436-441) -- against rdhip_rollout bit for bit and the oracle
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("tm", [False, True])
@pytest.mark.parametrize("shape", [(70, 9), (1, 2), (64, 17), (3, 1), (130, 33), (4101, 5)])
def test_rollout_split_equals_the_packed_rollout(shape, tm, dtype):
    """Every model family, both layouts, ragged groups and chunk ends (K = c C + 1: the last chunk holds no step), K = 1 (nothing to simulate), with and
    without x0: the simulated states equal those of rdhip_rollout's one-lane kernel on the packed stack BIT FOR BIT (the same step function; the kernel the
    dispatcher picks for rigid bodies integrates block by block along the cascade and differs in the last place), which the oracle pins."""
    M, K = shape
    rng = np.random.default_rng(M * 1000 + K)
    T = torch.float64 if dtype == np.float64 else torch.float32
    for model, om in ((RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0])), O.body()), (RD.Cartpole(), O.cartpole()), (RD.Quadrotor(bodyframe=True), O.quadrotor(frame=1)),
                      (RD.SatellitePair(), O.satellite_pair()), (RD.Satellite(R=RD.MRP), O.satellite(rot=O.ROT_MRP)), (RD.DoubleIntegrator(2), O.double_integrator(2))):
        n, m = RD.state_dim(model), RD.control_dim(model)
        nz = n + m
        data = rng.random((M, K, nz)) * 0.6
        if n == 13:
            q = rng.standard_normal((M, K, 4)); data[:, :, 3:7] = q / np.linalg.norm(q, axis=2, keepdims=True)
        if isinstance(model, RD.SatellitePair):
            for o in (0, 7):
                q = rng.standard_normal((M, K, 4)); data[:, :, o:o + 4] = q / np.linalg.norm(q, axis=2, keepdims=True)
        data = data.astype(dtype).astype(np.float64)
        dts = np.tile(rng.uniform(0.01, 0.05, K), (M, 1)); tt = np.cumsum(dts, axis=1) - dts
        if K > 4:
            dts[:, 3] = 0.0                                           # a terminal time step inside the trajectory: zero control for that step (src/knotpoint.jl:67)
        for rule, orule in ((RD.RK4, O.RK4), (RD.Euler, O.EULER)):
            fun = RD.DiscretizedDynamics(model, rule)
            for with_x0 in (False, True):
                x0 = (data[:, 0, :n] * 0.9).astype(dtype).astype(np.float64) if with_x0 else None
                if x0 is not None and n == 13:
                    x0[:, 3:7] = data[:, 0, 3:7]
                lay = (lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))) if tm else (lambda a: np.ascontiguousarray(a))
                st = {"data": torch.as_tensor(lay(data).astype(dtype)).cuda(), "t": torch.as_tensor(lay(tt)).cuda(), "dt": torch.as_tensor(lay(dts)).cuda()}
                x0d = None if x0 is None else torch.as_tensor(x0.astype(dtype)).cuda()
                RD.tune("rollout=plain")                              # the one-lane kernel: the same step function the split kernel's compute lane calls
                try:
                    RD.rollout_(RD.StaticReturn(), fun, st, x0=x0d, time_major=tm)
                finally:
                    RD.tune("reset")
                ref = st["data"][..., :n].contiguous()
                st2 = {"data": torch.as_tensor(lay(data).astype(dtype)).cuda(), "t": st["t"], "dt": st["dt"]}
                RD.rollout_(RD.StaticReturn(), fun, st2, x0=x0d, time_major=tm)         # the kernel the dispatcher picks (block by block along the cascade for rigid bodies: rounding-level differences)
                torch.testing.assert_close(st2["data"][..., :n].contiguous(), ref, rtol=1e-12 if dtype == np.float64 else 2e-5, atol=1e-13 if dtype == np.float64 else 2e-6)
                X = torch.full(lay(data[:, :, :n]).shape, float("nan"), dtype=T, device="cuda")
                if x0 is None:                                        # knot 0 is what X holds there
                    if tm:
                        X[0] = torch.as_tensor(data[:, 0, :n].astype(dtype)).cuda()
                    else:
                        X[:, 0] = torch.as_tensor(data[:, 0, :n].astype(dtype)).cuda()
                U = torch.as_tensor(lay(data[:, :K - 1, n:]).astype(dtype)).cuda().contiguous()
                RD.rollout_split_(RD.StaticReturn(), fun, X, U, st["t"], st["dt"], x0=x0d, time_major=tm)
                if K > 1:
                    assert "rollout_split_lane_kernel" in RD.last_kernel_name()
                assert torch.equal(X, ref), f"{type(model).__name__} {rule.__name__} M={M} K={K} tm={tm} x0={with_x0}: max diff {float((X - ref).abs().max())}"
                if M <= 130 and dtype == np.float64:
                    Zr = om.rollout(orule, data, tt, dts, X0=x0)
                    got = host(X).transpose(1, 0, 2) if tm else host(X)
                    np.testing.assert_allclose(got, Zr[:, :, :n], rtol=0, atol=TOL * max(1.0, np.abs(Zr).max()))


def test_rollout_split_host_arrays_and_errors():
    model = RD.Satellite()
    fun = RD.DiscretizedDynamics(model, RD.RK3)
    rng = np.random.default_rng(8)
    M, K = 5, 7
    data = H.rand_rigid(rng, M * K, 6).reshape(M, K, 19)
    tt = np.tile(np.arange(K) * 0.02, (M, 1)); dts = np.full((M, K), 0.02)
    X = np.full((M, K, 13), np.nan); X[:, 0] = data[:, 0, :13]
    U = np.ascontiguousarray(data[:, :K - 1, 13:])
    RD.rollout_split_(RD.StaticReturn(), fun, X, U, tt, dts)         # host arrays: staged through the context's workspace
    np.testing.assert_allclose(X, O.satellite().rollout(O.RK3, data, tt, dts)[:, :, :13], rtol=0, atol=TOL)
    with pytest.raises(ValueError):
        RD.rollout_split_(RD.StaticReturn(), fun, X, U[:, :-1], tt, dts)
    with pytest.raises(NotImplementedError):
        RD.rollout_split_(RD.StaticReturn(), RD.DiscretizedDynamics(model, RD.ImplicitMidpoint), X, U, tt, dts)


# ------------------------------------------------------------------------------------------------------------------------------------------
# the two-workgroups-per-CU forms of the fp32 one-launch kernels (csrc/fused.hpp fused_min_waves, csrc/fam_mixed.hip): what BASELINE config 5 runs on
# ------------------------------------------------------------------------------------------------------------------------------------------
def _tm_stack(rng, M, K, mctl, ulo, uhi, rigid=True, tm=True):
    Zn = H.rand_rigid(rng, M * K, mctl, ulo, uhi).reshape(M, K, 13 + mctl) if rigid else (rng.random((M, K, 5)) - np.array([0.4, 0.4, 0.4, 0.4, 0.0]))
    t = np.broadcast_to(np.arange(K) * 0.02, (M, K)).copy()
    dt = np.full((M, K), 0.02)
    dt[::7, K // 2] = 0.0                                             # a terminal knot inside some trajectories
    lay = (lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))) if tm else np.ascontiguousarray
    st = {"data": torch.as_tensor(lay(Zn).astype(np.float32)).cuda(), "t": torch.as_tensor(lay(t)).cuda(), "dt": torch.as_tensor(lay(dt)).cuda()}
    return Zn.astype(np.float32).astype(np.float64), t, dt, st


@pytest.mark.parametrize("tm", [True, False], ids=["time_major", "trajectory_major"])
@pytest.mark.parametrize("case", ["satellite_rk4", "quadrotor_rk3", "satellite_body_rk4"])
def test_two_per_cu_one_launch_form_ragged_groups_x0_and_the_one_per_cu_twin(case, tm):
    """rollout! + jacobian! in ONE launch for a batch that is bound by its Jacobian workgroups (65 groups of 64 trajectories, the last one ragged): fp32,
    time-major, an initial state of its own -- the kernel under 128 registers, two workgroups per CU.  States and Jacobians against the
    oracle, and bit for bit against the one-workgroup-per-CU kernel (RDHIP_TUNE fused_two_per_cu=0: the same arithmetic in another register budget) and
    against rdhip_rollout's states."""
    kind, rtag = case.rsplit("_", 1)
    rule, orule = (RD.RK4, O.RK4) if rtag == "rk4" else (RD.RK3, O.RK3)
    body = kind.endswith("_body")
    if kind.startswith("satellite"):
        model, om, mctl, ulo, uhi = RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0]), bodyframe=body), O.satellite(2.0, np.diag([2.0, 3.0, 1.0]), O.FRAME_BODY if body else O.FRAME_WORLD), 6, 0.0, 1.0
    else:
        model, om, mctl, ulo, uhi = RD.Quadrotor(), O.quadrotor(), 4, 0.6, 1.9
    rng = np.random.default_rng(606)
    M, K, n = 4100, 21, 13
    nz = n + mctl
    Zn, t, dt, st = _tm_stack(rng, M, K, mctl, ulo, uhi, tm=tm)
    x0 = Zn[:, 1, :n].copy(); x0[:, :3] *= 0.5
    x0 = x0.astype(np.float32)
    fun = RD.DiscretizedDynamics(model, rule)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    outs = {}
    for knob in ("fused_two_per_cu=1", "fused_two_per_cu=0"):
        s2 = {k: v.clone() for k, v in st.items()}
        J = RD.alloc_jacobians(fun, M * K, s2["data"]); J.fill_(float("nan"))
        RD.tune(knob)
        try:
            launches = RD.rollout_jacobian_(sig, AD, fun, J, s2, x0=torch.as_tensor(x0).cuda(), time_major=tm)
            name = RD.last_kernel_name()
        finally:
            RD.tune("reset")
        assert launches == 1 and "rollout_jac_fused_kernel" in name, (launches, name)
        outs[knob] = (s2["data"], J, name)
    assert outs["fused_two_per_cu=1"][2] != outs["fused_two_per_cu=0"][2]                  # two instantiations (the ring-depth argument carries the compact flag)
    assert torch.equal(outs["fused_two_per_cu=1"][0], outs["fused_two_per_cu=0"][0])
    assert torch.equal(outs["fused_two_per_cu=1"][1], outs["fused_two_per_cu=0"][1])
    s3 = {k: v.clone() for k, v in st.items()}
    RD.rollout_(sig, fun, s3, torch.as_tensor(x0).cuda(), time_major=tm)
    assert torch.equal(s3["data"], outs["fused_two_per_cu=1"][0])
    Zr = om.rollout(orule, Zn, t, dt, X0=x0.astype(np.float64))
    got = host(outs["fused_two_per_cu=1"][0]).astype(np.float64)
    got = got.transpose(1, 0, 2) if tm else got
    np.testing.assert_allclose(got[:, :, :n], Zr[:, :, :n], rtol=2e-4, atol=2e-4)
    np.testing.assert_array_equal(got[:, :, n:], Zn[:, :, n:])
    sl = slice(4032, 4100)                                            # the ragged last group and the one before it: Jacobians at the GPU's own fp32 states
    Zs = got[sl].reshape(-1, nz)
    Jo, _ = om.jacobian(orule, O.FORWARD_AD, Zs, t[sl].reshape(-1), dt[sl].reshape(-1))
    Jg = host(outs["fused_two_per_cu=1"][1]).astype(np.float64)
    Jg = (Jg.reshape(K, M, n, nz).transpose(1, 0, 2, 3) if tm else Jg.reshape(M, K, n, nz))[sl].reshape(-1, n, nz)
    np.testing.assert_allclose(Jg, Jo, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(Jo).max()))


@pytest.mark.parametrize("tm", [True, False], ids=["time_major", "trajectory_major"])
def test_two_per_cu_mixed_model_launch_ragged_groups(tm):
    """BASELINE config 5's call shape with ragged group counts: 4 100 satellite + 4 099 cartpole trajectories x 21 knots, fp32, ONE launch of the
    two-workgroups-per-CU mixed kernel; states of both models and the cartpole's Jacobians against the oracle, bit for bit against the one-per-CU twin."""
    rng = np.random.default_rng(607)
    MA, MB, K = 4100, 4099, 21
    sat, cart = RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0])), RD.Cartpole()
    Za, ta, da, sa = _tm_stack(rng, MA, K, 6, 0.0, 1.0, tm=tm)
    Zb, tb, db, sb = _tm_stack(rng, MB, K, 1, 0.0, 1.0, rigid=False, tm=tm)
    funs = [RD.DiscretizedDynamics(sat, RD.RK4), RD.DiscretizedDynamics(cart, RD.RK4)]
    outs = {}
    for knob in ("fused_two_per_cu=1", "fused_two_per_cu=0"):
        s1, s2 = {k: v.clone() for k, v in sa.items()}, {k: v.clone() for k, v in sb.items()}
        Js = [RD.alloc_jacobians(funs[0], MA * K, s1["data"]).fill_(float("nan")), RD.alloc_jacobians(funs[1], MB * K, s2["data"]).fill_(float("nan"))]
        RD.tune(knob)
        try:
            call = RD.PreparedRolloutJacobianMulti(RD.StaticReturn(), RD.ForwardAD(), funs, Js, [s1, s2], None, time_major=tm)
            call()
            assert call.launches == 1 and "rollout_jac_mixed_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
            name = RD.last_kernel_name()
        finally:
            RD.tune("reset")
        outs[knob] = (s1["data"], s2["data"], Js[0], Js[1], name)
    a, b = outs["fused_two_per_cu=1"], outs["fused_two_per_cu=0"]
    assert a[4] != b[4]
    for i in range(4):
        assert torch.equal(a[i], b[i]), i
    Zra = O.satellite(2.0, np.diag([2.0, 3.0, 1.0])).rollout(O.RK4, Za, ta, da)
    Zrb = O.cartpole().rollout(O.RK4, Zb, tb, db)
    ga = host(a[0]).astype(np.float64)
    np.testing.assert_allclose((ga.transpose(1, 0, 2) if tm else ga)[:, :, :13], Zra[:, :, :13], rtol=2e-4, atol=2e-4)
    gb = host(a[1]).astype(np.float64)
    gb = gb.transpose(1, 0, 2) if tm else gb
    np.testing.assert_allclose(gb[:, :, :4], Zrb[:, :, :4], rtol=2e-4, atol=2e-4)
    sl = slice(4032, 4099)
    Jo, _ = O.cartpole().jacobian(O.RK4, O.FORWARD_AD, gb[sl].reshape(-1, 5), tb[sl].reshape(-1), db[sl].reshape(-1))
    Jg = host(a[3]).astype(np.float64)
    Jg = (Jg.reshape(K, MB, 4, 5).transpose(1, 0, 2, 3) if tm else Jg.reshape(MB, K, 4, 5))[sl].reshape(-1, 4, 5)
    np.testing.assert_allclose(Jg, Jo, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(Jo).max()))
