"""GPU tests added in round 5: the fused rollout -> Jacobians launch after its re-organisation (one walk instead of two phases, cached progress words,
back-off), the pipelined rollout / Jacobian pieces of models without the fused launch, dispatch by kernel IDENTITY for every entry point and for the
thresholds of the dynamic tile schedule, and the NaN / Inf contract of include/rdhip.h."""
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


def host(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)


def _tm_stack(data, t, dt, cast=lambda a: a):
    """(M, K, nz) host arrays -> the time-major device stack {"data" (K, M, nz), "t", "dt" (K, M)}"""
    return {"data": torch.as_tensor(cast(np.ascontiguousarray(data.transpose(1, 0, 2)))).cuda(),
            "t": torch.as_tensor(np.ascontiguousarray(t.T)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T)).cuda()}


def _flat(model, st, K, M):
    n, m = RD.state_dim(model), RD.control_dim(model)
    return RD.PackedTrajectory(n, m, st["data"].reshape(K * M, n + m), st["t"].reshape(-1), st["dt"].reshape(-1))


# ------------------------------------------------------------------------------------------------------------------------------------------
# rollout! + jacobian! in pieces (models without the fused launch): src/trajectories.jl:436-441 + src/functionbase.jl:242-257
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("K", [256, 101])
def test_pipelined_rollout_jacobian_equals_the_two_plain_launches(dtype, K):
    """A time-major cartpole stack of bandwidth size is rolled out in four pieces with the Jacobians of a finished piece on a second stream
    (rdhip.h: rdhip_rollout_jacobian).  States must be bit-identical to rdhip_rollout on the whole stack (a piece restarts from the state the piece
    before it stored), Jacobians bit-identical to rdhip_jacobian on the simulated stack, and both agree with the oracle."""
    rng = np.random.default_rng(17)
    M = 4096
    model, om = RD.Cartpole(), O.cartpole()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    data = 0.2 * rng.random((M, K, 5)); t = np.tile(np.arange(K) * 0.01, (M, 1)); dt = np.full((M, K), 0.01)
    dt[:, -1] = 0.0                                        # a terminal last knot point (dt === 0: controls of that knot do not count)
    cast = (lambda a: a.astype(dtype))
    st = _tm_stack(data, t, dt, cast)
    J = RD.alloc_jacobians(dm, K * M, st["data"]).fill_(7.0)
    launches = RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, st, None, time_major=True)
    assert launches == 8, launches                          # four rollout pieces + four Jacobian pieces
    torch.cuda.synchronize()
    st2 = _tm_stack(data, t, dt, cast)
    RD.rollout_(RD.StaticReturn(), dm, st2, None, time_major=True)
    assert torch.equal(st["data"], st2["data"])
    J2 = RD.alloc_jacobians(dm, K * M, st2["data"])
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J2, None, _flat(model, st2, K, M))
    assert torch.equal(J, J2)
    sample = [0, 63, 64, 4095]
    ref = om.rollout(O.RK4, cast(data[sample]).astype(np.float64), t[sample], dt[sample])
    tol = 1e-10 if dtype == np.float64 else 2e-3
    np.testing.assert_allclose(host(st["data"])[:, sample].transpose(1, 0, 2)[..., :4], ref[..., :4], rtol=tol, atol=tol)
    Jg = host(J).reshape(K, M, 4, 5)
    rolled = host(st["data"])
    for j in sample:
        Jo, _ = om.jacobian(O.RK4, O.FORWARD_AD, rolled[:, j].astype(np.float64), t[j], dt[j])
        np.testing.assert_allclose(Jg[:, j], Jo, rtol=0, atol=1e-10 if dtype == np.float64 else 2e-4)


def test_a_rollout_that_fills_the_chip_is_not_cut_into_pieces():
    """16 384 trajectories are 256 rollout workgroups: nothing is idle beside them, the Jacobians of a finished piece would have nowhere to run -- the
    call issues its two plain launches (measured: 489 us in pieces against 390 us for 16 384 x 64 fp64 satellite knots)."""
    rng = np.random.default_rng(19)
    M, K = 16384, 64
    dm = RD.DiscretizedDynamics(RD.Cartpole(), RD.RK4)
    data = (0.2 * rng.random((M, K, 5))).astype(np.float32); t = np.tile(np.arange(K) * 0.01, (M, 1)); dt = np.full((M, K), 0.01)
    st = _tm_stack(data, t, dt)
    J = RD.alloc_jacobians(dm, K * M, st["data"])
    assert RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, st, None, time_major=True) == 2
    J2 = RD.alloc_jacobians(dm, K * M, st["data"])
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J2, None, _flat(RD.Cartpole(), st, K, M))
    torch.cuda.synchronize()
    assert torch.equal(J, J2)


def test_pipelined_pieces_of_a_custom_model_and_a_second_call_on_the_same_context():
    """The pieces take any model that has no fused launch -- here a hipRTC cartpole -- and the internal stream is joined at the end of every call: a
    second call that reuses the arrays right away sees the first one's results."""
    rng = np.random.default_rng(18)
    M, K = 2048, 160
    model = RD.CustomModel(4, 1, H.CARTPOLE_SRC, params=[1.0, 0.2, 0.5, 9.81])
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    data = 0.2 * rng.random((M, K, 5)); t = np.tile(np.arange(K) * 0.01, (M, 1)); dt = np.full((M, K), 0.01)
    st = _tm_stack(data, t, dt)
    J = RD.alloc_jacobians(dm, K * M, st["data"])
    assert RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, st, None, time_major=True) == 8
    first = J.clone()                                       # (ordered on the caller's stream behind the call)
    assert RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, st, None, time_major=True) == 8
    torch.cuda.synchronize()
    assert torch.equal(first, J)                            # the rollout of simulated states reproduces them: same inputs, same bits
    om = O.cartpole()
    rolled = host(st["data"])
    Jo, _ = om.jacobian(O.RK4, O.FORWARD_AD, rolled[:, 5], t[5], dt[5])
    np.testing.assert_allclose(host(J).reshape(K, M, 4, 5)[:, 5], Jo, rtol=0, atol=TOL)


# ------------------------------------------------------------------------------------------------------------------------------------------
# the fused launch: one walk through both parts, cached progress words
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,dtype", [(4096, 64, np.float32), (1024, 96, np.float32), (192, 40, np.float64), (4096, 33, np.float64), (132, 37, np.float32)])
def test_fused_rollout_jacobians_walk(M, K, dtype):
    """rdhip_rollout_jacobian, satellite, time-major, ONE launch: with 64 groups the fp32 launch walks both parts (the Jacobian workgroups' share and
    the part every workgroup joins), with fewer it has one part; fp64 never has a second part.  Jacobians must equal rdhip_jacobian on the simulated
    stack to rounding and the states the plain rollout bit for bit."""
    rng = np.random.default_rng(19)
    model, om = RD.Satellite(), O.satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    data = 0.3 * H.rand_rigid(rng, M * K, 6).reshape(M, K, 19) + np.array([0] * 3 + [0.7, 0, 0, 0] + [0] * 12)
    t = np.tile(np.arange(K) * 0.01, (M, 1)); dt = np.full((M, K), 0.01)
    cast = (lambda a: a.astype(dtype))
    st = _tm_stack(data, t, dt, cast)
    J = RD.alloc_jacobians(dm, K * M, st["data"]).fill_(7.0)
    assert RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, st, None, time_major=True) == 1
    assert "rollout_jac_fused_kernel" in RD.last_kernel_name()
    torch.cuda.synchronize()
    st2 = _tm_stack(data, t, dt, cast)
    RD.rollout_(RD.StaticReturn(), dm, st2, None, time_major=True)
    assert torch.equal(st["data"], st2["data"])
    J2 = RD.alloc_jacobians(dm, K * M, st2["data"])
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J2, None, _flat(model, st2, K, M))
    d = (J - J2).abs().max().item()
    assert d <= (1e-14 if dtype == np.float64 else 2e-6) * max(1.0, J2.abs().max().item()), d
    j = M - 1
    Jo, _ = om.jacobian(O.RK4, O.FORWARD_AD, host(st["data"])[:, j].astype(np.float64), t[j], dt[j])
    np.testing.assert_allclose(host(J).reshape(K, M, 13, 19)[:, j], Jo, rtol=0, atol=TOL if dtype == np.float64 else 2e-4)


# ------------------------------------------------------------------------------------------------------------------------------------------
# dispatch by identity: every entry point, and both sides of the thresholds of the dynamic tile schedule
# ------------------------------------------------------------------------------------------------------------------------------------------
def _is_dyn(name):
    """the launch took the dynamic tile schedule: rdhip_last_kernel_name appends a marker (round 6: only the ForwardAD flagship keeps a schedule-free
    instantiation of its own -- `..., 4, true>(` in the name --, every other body carries the schedule in its one instantiation and takes it at run time)"""
    return "[dynamic tile schedule]" in name


def _sat_traj(rng, N, dtype=np.float64, model=None):
    model = model or RD.Satellite()
    m = RD.control_dim(model)
    return RD.PackedTrajectory(13, m, np.ascontiguousarray(H.rand_rigid(rng, N, m), dtype=dtype), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda")


def test_dynamic_tile_schedule_thresholds_by_kernel_name():
    """The instantiation with the dynamic tile schedule (template argument DYN = true: the demangled name ends in `, true>`) must be the one that runs from
    8 tiles per workgroup of the reduced grid on -- 3/4 of the compute units, dispatch.hpp dyn_grid -- for fp64, and never for fp32; one size class
    further up (128 tiles per workgroup) the per-XCD-range counters are the same instantiation.  (Commit 5efa8e2 of round 4: the selection test never
    fired and every parity test stayed green.)"""
    rng = np.random.default_rng(23)
    sig, AD, FD, UD = RD.StaticReturn(), RD.ForwardAD(), RD.FiniteDifference(), RD.UserDefined()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    grid = (cus * 3 // 4) & ~7
    n_dyn = 8 * grid * 64                                    # the first size with 8 tiles per workgroup
    dm = RD.DiscretizedDynamics(RD.Satellite(), RD.RK4)
    for N, dtype, want_dyn in [(n_dyn - 64, np.float64, False), (n_dyn, np.float64, True), (n_dyn + 12345, np.float64, True), (n_dyn, np.float32, False),
                               (65536, np.float64, False)]:
        Z = _sat_traj(rng, N, dtype)
        J = RD.alloc_jacobians(dm, N, Z.data)
        for diff, kern in ((AD, "jac_ad_split_kernel"), (FD, "jac_fd_cascade_kernel"), (UD, "jac_chain_split_kernel")):
            RD.jacobian_(sig, diff, dm, J, None, Z)
            name = RD.last_kernel_name()
            assert kern in name and _is_dyn(name) == want_dyn, (N, dtype.__name__, type(diff).__name__, name)
        if dtype == np.float64:
            E = RD.alloc_error_expansions(RD.Satellite(), N, Z.data)
            RD.error_expansion_(dm, E, None, Z)
            name = RD.last_kernel_name()
            assert "jac_errexp_split_kernel" in name and _is_dyn(name) == ((N - 1 + 63) // 64 >= 8 * grid), (N, name)
        del Z, J
    torch.cuda.synchronize()
    # the dynamic launch gives the static launch's bits (a big batch against its two halves)
    N = n_dyn + 64 * 100
    Z = _sat_traj(rng, N)
    J = RD.alloc_jacobians(dm, N, Z.data)
    RD.jacobian_(sig, AD, dm, J, None, Z)
    h = N // 2 // 64 * 64
    for lo, hi in ((0, h), (h, N)):
        Zp = RD.PackedTrajectory(13, 6, Z.data[lo:hi], Z.t[lo:hi], Z.dt[lo:hi])
        Jp = RD.alloc_jacobians(dm, hi - lo, Z.data)
        RD.jacobian_(sig, AD, dm, Jp, None, Zp)
        assert not _is_dyn(RD.last_kernel_name())
        assert torch.equal(J[lo:hi], Jp)


def test_dispatch_identity_of_the_other_entry_points():
    """rdhip_jacobian_multi (inline table / device table / uniform batches), rdhip_rollout[_layout] (one lane per trajectory / cascade / streaming, by
    model and size), rdhip_rollout_jacobian (one launch / pieces / two launches), rdhip_dynamics_error_jacobian, rdhip_error_expansion,
    rdhip_errstate_jacobian in both modes, rdhip_state_diff: each must reach the kernel DESIGN.md names for it."""
    rng = np.random.default_rng(29)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    sat, cart, quad = RD.Satellite(), RD.Cartpole(), RD.Quadrotor()
    dsat, dcart = RD.DiscretizedDynamics(sat, RD.RK4), RD.DiscretizedDynamics(cart, RD.RK4)

    def multi(dm, sizes, dtype=np.float64):
        model = dm.continuous_dynamics if hasattr(dm, "continuous_dynamics") else None
        n, m = RD.state_dim(dm), RD.control_dim(dm)
        Zs, Js = [], []
        for N in sizes:
            data = H.rand_rigid(rng, N, m) if n == 13 else 0.2 * rng.random((N, n + m))
            Zs.append(RD.PackedTrajectory(n, m, np.ascontiguousarray(data, dtype=dtype), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda"))
            Js.append(RD.alloc_jacobians(dm, N, Zs[-1].data))
        RD.jacobian_multi_(sig, AD, dm, Js, None, Zs)
        return RD.last_kernel_name()

    assert "jac_split_multi_kernel" in multi(dsat, [1024] * 16)                    # inline table (<= 32 batches)
    assert "jac_split_multi_kernel" in multi(dsat, [101] * 200)                    # device table, uniform batches
    assert "jac_split_multi_kernel" in multi(dsat, [101, 77, 300] * 30)            # device table, ragged
    name = multi(dsat, [4096 * 8] * 64)                                            # 32 768 tiles: the dynamic schedule of long trajectories
    assert "jac_split_multi_kernel" in name and _is_dyn(name), name
    assert "jac_ad_small_multi_kernel" in multi(dcart, [4096] * 16)
    torch.cuda.synchronize()

    def stack(model, M, K, dtype=np.float64, time_major=False):
        n, m = RD.state_dim(model), RD.control_dim(model)
        data = (H.rand_rigid(rng, M * K, m) if n == 13 else 0.2 * rng.random((M * K, n + m))).reshape(M, K, n + m).astype(dtype)
        t = np.tile(np.arange(K) * 0.01, (M, 1)); dt = np.full((M, K), 0.01)
        if time_major:
            return _tm_stack(data, t, dt)
        return {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(t).cuda(), "dt": torch.as_tensor(dt).cuda()}

    RD.rollout_(sig, dcart, stack(cart, 512, 64))
    assert "rollout_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    RD.rollout_(sig, dsat, stack(sat, 512, 64))
    assert "rollout_jac_fused_kernel" in RD.last_kernel_name() or "rollout_cascade_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    RD.rollout_(sig, RD.DiscretizedDynamics(RD.Quadrotor(bodyframe=True), RD.RK4), stack(quad, 64 * 200, 8))      # more groups than half the chip, fewer than two per CU: one lane per trajectory
    assert "rollout_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    RD.rollout_(sig, RD.DiscretizedDynamics(RD.Quadrotor(J=np.array([[0.0023, 1e-4, 0], [1e-4, 0.0023, 0], [0, 0, 0.004]])), RD.RK4), stack(quad, 512, 64))      # a lean variant (full inertia): the cascade kernel
    assert "rollout_cascade_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    RD.rollout_(sig, dsat, stack(sat, 32768, 16, time_major=True), time_major=True)  # two groups per CU and more: the streaming kernel
    assert "rollout_stream_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    torch.cuda.synchronize()

    st = stack(sat, 512, 64, time_major=True)
    J = RD.alloc_jacobians(dsat, 512 * 64, st["data"])
    assert RD.rollout_jacobian_(sig, AD, dsat, J, st, None, time_major=True) == 1 and "rollout_jac_fused_kernel" in RD.last_kernel_name()
    st = stack(sat, 512, 64)
    assert RD.rollout_jacobian_(sig, AD, dsat, J, st, None) == 1 and "rollout_jac_fused_kernel" in RD.last_kernel_name()      # trajectory-major, latency-sized: gathered tiles
    st = stack(sat, 4096, 80)
    J = RD.alloc_jacobians(dsat, 4096 * 80, st["data"])
    assert RD.rollout_jacobian_(sig, AD, dsat, J, st, None) == 2 and "jac_ad_split_kernel" in RD.last_kernel_name()           # ... bandwidth-sized: two launches
    st = stack(cart, 4096, 64, time_major=True)
    J = RD.alloc_jacobians(dcart, 4096 * 64, st["data"])
    assert RD.rollout_jacobian_(sig, AD, dcart, J, st, None, time_major=True) == 8 and "jac_ad_small_kernel" in RD.last_kernel_name()   # pieces
    st = stack(cart, 512, 64, time_major=True)
    J = RD.alloc_jacobians(dcart, 512 * 64, st["data"])
    assert RD.rollout_jacobian_(sig, AD, dcart, J, st, None, time_major=True) == 2 and "jac_ad_small_kernel" in RD.last_kernel_name()
    torch.cuda.synchronize()

    N = 1 << 16
    Z = _sat_traj(rng, N)
    J1 = RD.alloc_jacobians(dsat, N - 1, Z.data); J2 = RD.alloc_jacobians(dsat, N - 1, Z.data)
    RD.dynamics_error_jacobian_(sig, AD, dsat, J2, J1, None, None, Z)
    assert "fill_neg_identity_kernel" in RD.last_kernel_name() or "jac_ad_split_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    E = RD.alloc_error_expansions(sat, N, Z.data)
    RD.error_expansion_(dsat, E, None, Z)
    assert "jac_errexp_split_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    X = torch.as_tensor(H.rand_rigid(rng, 1 << 18, 4)[:, :13]).cuda()
    G = RD.alloc_errstate_jacobians(quad, 1 << 18, X)
    RD.errstate_jacobian_(quad, G, X, clear=True)
    assert "errstate_tile_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    RD.errstate_jacobian_(quad, G, X)
    assert "errstate_pattern_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    dX = torch.empty((1 << 18, 12), dtype=torch.float64, device="cuda")
    RD.state_diff_(quad, dX, X, X.flip(0).contiguous())
    assert "state_diff_tile_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------------------------------------------------
# NaN / Inf contract (include/rdhip.h): the reference creates trajectories filled with NaN (src/trajectories.jl:83) and stores dt = Inf on a last
# knot point that carries a control (:104-108)
# ------------------------------------------------------------------------------------------------------------------------------------------
def _poisoned(rng, N, bad):
    Zn = H.rand_rigid(rng, N, 6)
    for k, comp in bad:
        Zn[k, comp] = np.nan
    return Zn


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_nan_knot_points_stay_nan_and_their_neighbours_exact(dtype):
    """A knot point with a NaN component: every output entry of THAT knot point that depends on the component is NaN (entries that are structurally
    zero stay 0: the tangent seeds are compile-time constants, rdhip.h), the call returns, and every other knot point of the tile -- neighbours in the
    same 64-knot tile included -- gets exactly what it gets without the NaN."""
    rng = np.random.default_rng(31)
    N = 1000
    sig, AD, FD = RD.StaticReturn(), RD.ForwardAD(), RD.FiniteDifference()
    sat = RD.Satellite()
    dm = RD.DiscretizedDynamics(sat, RD.RK4)
    bad = [(5, 10), (70, 4), (130, 15), (999, 0)]               # w_x, a quaternion component, a force control, a position
    clean = H.rand_rigid(rng, N, 6)
    dirty = clean.copy()
    for k, comp in bad:
        dirty[k, comp] = np.nan
    t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
    badk = [k for k, _ in bad]
    good = np.setdiff1d(np.arange(N), badk)
    for diff in (AD, FD):
        outs = []
        for Zn in (clean, dirty):
            Z = RD.PackedTrajectory(13, 6, np.ascontiguousarray(Zn, dtype=dtype), t, dt).to("cuda")
            J = RD.alloc_jacobians(dm, N, Z.data); Y = torch.empty((N, 13), dtype=Z.data.dtype, device="cuda")
            RD.jacobian_(sig, diff, dm, J, Y, Z)
            torch.cuda.synchronize()
            outs.append((host(J), host(Y)))
        (Jc, Yc), (Jd, Yd) = outs
        np.testing.assert_array_equal(Jd[good], Jc[good]); np.testing.assert_array_equal(Yd[good], Yc[good])
        assert np.isfinite(Jc).all() and np.isfinite(Yc).all()
        for k, comp in bad:
            assert np.isnan(Yd[k]).any(), (k, comp)
            # Jacobian entries of a poisoned knot point: NaN wherever the entry's formula reads the poisoned input (the w and q poisons: the dynamics are
            # non-linear in both), possibly finite elsewhere -- tangent seeds are compile-time constants, so the `0 * NaN` terms of a ForwardDiff
            # evaluation do not exist here (the force-control and position poisons: the step is linear in u and r, their columns never read them).
            if comp in (10, 4):
                assert np.isnan(Jd[k][:, comp]).any(), (type(diff).__name__, k, comp)
    # discrete dynamics, continuous dynamics
    for fun in (dm, sat):
        Z = RD.PackedTrajectory(13, 6, np.ascontiguousarray(dirty, dtype=dtype), t, dt).to("cuda")
        Y = torch.empty((N, 13), dtype=Z.data.dtype, device="cuda")
        RD.evaluate_(fun, Y, Z)
        Zc = RD.PackedTrajectory(13, 6, np.ascontiguousarray(clean, dtype=dtype), t, dt).to("cuda")
        Yc = torch.empty((N, 13), dtype=Z.data.dtype, device="cuda")
        RD.evaluate_(fun, Yc, Zc)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(host(Y)[good], host(Yc)[good])
        for k, comp in bad:
            if comp >= 3 or fun is dm:                      # (continuous dynamics do not read the position)
                assert np.isnan(host(Y)[k]).any(), (k, comp)
    # errstate_jacobian!, state_diff: the NaN quaternion
    X = torch.as_tensor(np.ascontiguousarray(dirty[:, :13], dtype=dtype)).cuda()
    Xc = torch.as_tensor(np.ascontiguousarray(clean[:, :13], dtype=dtype)).cuda()
    G = RD.alloc_errstate_jacobians(sat, N, X); Gc = RD.alloc_errstate_jacobians(sat, N, X)
    RD.errstate_jacobian_(sat, G, X, clear=True); RD.errstate_jacobian_(sat, Gc, Xc, clear=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(G)[good], host(Gc)[good])
    assert np.isnan(host(G)[70, 3:7, 3:6]).any() and np.isfinite(host(G)[5]).all()


def test_nan_and_inf_through_the_implicit_midpoint_newton_solve_and_the_rollout():
    """ImplicitMidpoint: Newton on a NaN knot point must TERMINATE (the iteration limit ends it) and leave NaN; dt = Inf on the last knot point of a
    trajectory (the reference's convention for a last knot that carries a control) must not disturb the knots in front of it; a rollout that starts
    from a NaN state is NaN from there on and leaves the other trajectories alone."""
    rng = np.random.default_rng(37)
    N = 640
    sat = RD.Satellite()
    dmi = RD.DiscretizedDynamics(sat, RD.ImplicitMidpoint)
    ift = RD.ImplicitFunctionTheorem(RD.ForwardAD())
    clean = 0.3 * H.rand_rigid(rng, N, 6) + np.array([0] * 3 + [0.7, 0, 0, 0] + [0] * 12)
    dirty = clean.copy(); dirty[77, 11] = np.nan
    t, dt = np.arange(N) * 0.05, np.full(N, 0.05)
    dtinf = dt.copy(); dtinf[-1] = np.inf
    res = {}
    for tag, Zn, h in (("clean", clean, dt), ("nan", dirty, dt), ("inf", clean, dtinf)):
        Z = RD.PackedTrajectory(13, 6, Zn, t, h).to("cuda")
        J = RD.alloc_jacobians(dmi, N, Z.data); Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
        RD.jacobian_(RD.StaticReturn(), ift, dmi, J, Y, Z)
        torch.cuda.synchronize()                             # (a Newton loop that never ended would hang here: the test has a time limit)
        res[tag] = (host(J), host(Y))
    good = np.setdiff1d(np.arange(N), [77])
    # (neighbours in the poisoned knot's 64-knot tile are redone by the exact LDS kernel, whose arithmetic differs from the register kernel's by rounding)
    np.testing.assert_allclose(res["nan"][0][good], res["clean"][0][good], rtol=0, atol=1e-13); np.testing.assert_allclose(res["nan"][1][good], res["clean"][1][good], rtol=0, atol=1e-13)
    far = good[(good < 64) | (good >= 128)]
    np.testing.assert_array_equal(res["nan"][0][far], res["clean"][0][far]); np.testing.assert_array_equal(res["nan"][1][far], res["clean"][1][far])
    assert np.isnan(res["nan"][1][77]).any() and np.isnan(res["nan"][0][77]).any()
    np.testing.assert_allclose(res["inf"][0][:-1], res["clean"][0][:-1], rtol=0, atol=1e-13); np.testing.assert_allclose(res["inf"][1][:-1], res["clean"][1][:-1], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(res["inf"][0][:576], res["clean"][0][:576]); np.testing.assert_array_equal(res["inf"][1][:576], res["clean"][1][:576])      # (tiles other than the last knot's)
    # explicit rule, dt = Inf on the last knot
    dm = RD.DiscretizedDynamics(sat, RD.RK4)
    outs = []
    for h in (dt, dtinf):
        Z = RD.PackedTrajectory(13, 6, clean, t, h).to("cuda")
        J = RD.alloc_jacobians(dm, N, Z.data)
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, None, Z)
        torch.cuda.synchronize()
        outs.append(host(J))
    np.testing.assert_array_equal(outs[0][:-1], outs[1][:-1])
    # rollout: trajectory 3 starts from a NaN state
    M, K = 130, 40
    data = (0.3 * H.rand_rigid(rng, M * K, 6) + np.array([0] * 3 + [0.7, 0, 0, 0] + [0] * 12)).reshape(M, K, 19)
    tt = np.tile(np.arange(K) * 0.01, (M, 1)); hh = np.full((M, K), 0.01)
    st = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(hh).cuda()}
    RD.rollout_(RD.StaticReturn(), dm, st)
    d2 = data.copy(); d2[3, 0, 8] = np.nan
    st2 = {"data": torch.as_tensor(d2).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(hh).cuda()}
    RD.rollout_(RD.StaticReturn(), dm, st2)
    torch.cuda.synchronize()
    a, b = host(st["data"]), host(st2["data"])
    others = np.setdiff1d(np.arange(M), [3])
    np.testing.assert_array_equal(a[others], b[others])
    assert np.isnan(b[3, 1:, :13]).any(axis=1).all()          # every simulated knot of the poisoned trajectory carries a NaN


# ------------------------------------------------------------------------------------------------------------------------------------------
# plug-in: user-defined DISCRETE models and user-written Jacobians (src/discrete_dynamics.jl:1-113, src/dynamics.jl:115-116)
# ------------------------------------------------------------------------------------------------------------------------------------------
DI_STEP = "xn[0] = x[0] + x[2] * h; xn[1] = x[1] + x[3] * h; xn[2] = x[2] + u[0] * h; xn[3] = x[3] + u[1] * h;"          # test/discrete_dynamics_test.jl:35-42
DI_JAC = "J[0] = 1; J[8] = h; J[5] = 1; J[13] = h; J[10] = 1; J[18] = h; J[15] = 1; J[23] = h;"                          # :52-63 (J[i + 4 j])
# test/cartpole_model.jl:57-96 restated on plain arrays: H qdd = -(C qd + G - B u); d qdd / d z = H \ (-dH/dz qdd - dC/dz qd - dG/dz + dB/dz)
CART_JAC = """
    const T mc = p[0], mp = p[1], l = p[2], g = p[3];
    const T s = sin(x[1]), c = cos(x[1]);
    const T H00 = mc + mp, H01 = mp * l * c, H11 = mp * l * l;
    const T det = H00 * H11 - H01 * H01;
    const T r0 = -mp * x[3] * l * s * x[3] - u[0], r1 = mp * g * l * s;
    const T qdd0 = -(H11 * r0 - H01 * r1) / det, qdd1 = -(-H01 * r0 + H00 * r1) / det;
    T M0[5] = {0, mp * l * s * qdd1 + mp * l * c * x[3] * x[3], 0, 2 * mp * l * x[3] * s, 1};
    T M1[5] = {0, mp * l * s * qdd0 - mp * g * l * c, 0, 0, 0};
    J[0 + 4 * 2] = 1; J[1 + 4 * 3] = 1;
    for (int j = 0; j < 5; ++j) { J[2 + 4 * j] = (H11 * M0[j] - H01 * M1[j]) / det; J[3 + 4 * j] = (-H01 * M0[j] + H00 * M1[j]) / det; }
"""


def _di_closed_form(N, dt):
    J = np.zeros((N, 4, 6))
    for i in range(4):
        J[:, i, i] = 1.0
    J[:, 0, 2] = dt; J[:, 1, 3] = dt; J[:, 2, 4] = dt; J[:, 3, 5] = dt
    return J


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_user_defined_discrete_model_double_integrator(dtype):
    """The reference's discrete double integrator (test/discrete_dynamics_test.jl:5-63) as a plug-in model: discrete_dynamics, jacobian! with the three
    diff methods against the closed form of :52-63 (ForwardAD and UserDefined exact, FiniteDifference to 1e-6 as test/integration_tests.jl:14 asks),
    dynamics_error and dynamics_error_jacobian! (J2 = [-I 0], :65-73), rollout!, the terminal-knot rule, and rule checking."""
    rng = np.random.default_rng(41)
    N = 1000
    di = RD.CustomDiscreteModel(4, 2, DI_STEP, jacobian=DI_JAC)
    assert RD.dims(di) == (4, 2, 4) and isinstance(RD.default_diffmethod(di), RD.ForwardAD)
    Zn = rng.random((N, 6)).astype(dtype); t = np.arange(N) * 0.1; dt = 0.05 + 0.1 * rng.random(N)
    dt[-1] = 0.0
    Z = RD.PackedTrajectory(4, 2, Zn, t, dt).to("cuda")
    Y = torch.empty((N, 4), dtype=Z.data.dtype, device="cuda")
    RD.evaluate_(di, Y, Z)
    h = dt.astype(dtype)[:, None]
    u = Zn[:, 4:6].copy(); u[-1] = 0                          # terminal knot point: control(z) = 0
    ref = np.concatenate([Zn[:, 0:2] + Zn[:, 2:4] * h, Zn[:, 2:4] + u * h], axis=1)
    np.testing.assert_allclose(host(Y), ref, rtol=0, atol=1e-15 if dtype == np.float64 else 1e-6)
    Jref = _di_closed_form(N, dt)
    Jref[-1, :, 4:] = 0.0                                     # control columns of a terminal knot point
    sig = RD.StaticReturn()
    for diff, tol in ((RD.ForwardAD(), 1e-15), (RD.UserDefined(), 1e-15), (RD.FiniteDifference(), 1e-6)):
        J = RD.alloc_jacobians(di, N, Z.data).fill_(7.0)
        Y.fill_(7.0)
        RD.jacobian_(sig, diff, di, J, Y, Z)
        np.testing.assert_allclose(host(J), Jref, rtol=0, atol=tol if dtype == np.float64 else max(tol, 2e-3 if isinstance(diff, RD.FiniteDifference) else 1e-6))
        np.testing.assert_allclose(host(Y), ref, rtol=0, atol=1e-15 if dtype == np.float64 else 1e-6)
    assert "jac_user_kernel" in RD.last_kernel_name() or "jac_fd" in RD.last_kernel_name()
    if dtype == np.float64:
        J1 = RD.alloc_jacobians(di, N - 1, Z.data); J2 = RD.alloc_jacobians(di, N - 1, Z.data)
        RD.dynamics_error_jacobian_(sig, RD.UserDefined(), di, J2, J1, None, None, Z)
        np.testing.assert_allclose(host(J1), Jref[:-1], rtol=0, atol=1e-15)
        J2r = np.zeros((N - 1, 4, 6)); J2r[:, range(4), range(4)] = -1.0
        np.testing.assert_array_equal(host(J2), J2r)
        e = RD.dynamics_error(di, Z)
        np.testing.assert_allclose(host(e), ref[:-1] - Zn[1:, :4], rtol=0, atol=1e-15)
        # rollout!: x_{k+1} = step(x_k, u_k)
        M, K = 70, 30
        data = rng.random((M, K, 6)); tt = np.tile(np.arange(K) * 0.1, (M, 1)); hh = np.full((M, K), 0.1)
        st = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(hh).cuda()}
        RD.rollout_(sig, di, st)
        x = data[:, 0, :4].copy(); out = host(st["data"])
        for k in range(K - 1):
            x = np.concatenate([x[:, 0:2] + x[:, 2:4] * 0.1, x[:, 2:4] + data[:, k, 4:6] * 0.1], axis=1)
            np.testing.assert_allclose(out[:, k + 1, :4], x, rtol=0, atol=1e-13)
    # a discrete model has no quadrature rule and no continuous dynamics; without a Jacobian body UserDefined is NotImplemented like the reference
    with pytest.raises((ValueError, TypeError, NotImplementedError, RD.RdhipError)):
        RD.dynamics(di, Z)
    bare = RD.CustomDiscreteModel(4, 2, DI_STEP)
    J = RD.alloc_jacobians(bare, N, Z.data)
    with pytest.raises(NotImplementedError):
        RD.jacobian_(sig, RD.UserDefined(), bare, J, None, Z)
    RD.jacobian_(sig, RD.ForwardAD(), bare, J, None, Z)
    np.testing.assert_allclose(host(J), Jref, rtol=0, atol=1e-15 if dtype == np.float64 else 1e-6)


@pytest.mark.parametrize("rule", [None, RD.Euler, RD.RK2, RD.RK3, RD.RK4])
def test_user_written_continuous_jacobian_of_the_cartpole(rule):
    """The cartpole's hand-written jacobian! (test/cartpole_model.jl:57-96) as the plug-in's Jacobian body: UserDefined() evaluates those formulas --
    directly for the continuous model, through the integrators' chain rule on the stage Jacobians for DiscretizedDynamics -- and must agree with the
    oracle's UserDefined (the same formulas in matrix form, oracle/rd_oracle.cpp user_jacobian + chain rule) to 1e-10, and with ForwardAD to 1e-10 as
    test/integration_tests.jl:13-17 asks of the reference."""
    rng = np.random.default_rng(43)
    N = 777
    cart = RD.CustomModel(4, 1, H.CARTPOLE_SRC, params=[1.0, 0.2, 0.5, 9.81], jacobian=CART_JAC)
    fun = cart if rule is None else RD.DiscretizedDynamics(cart, rule)
    Zn = rng.random((N, 5)) - 0.3; t = np.arange(N) * 0.1; dt = np.full(N, 0.1)
    Z = RD.PackedTrajectory(4, 1, Zn, t, dt).to("cuda")
    sig = RD.StaticReturn()
    J = RD.alloc_jacobians(fun, N, Z.data).fill_(7.0); Y = torch.empty((N, 4), dtype=torch.float64, device="cuda")
    RD.jacobian_(sig, RD.UserDefined(), fun, J, Y, Z)
    assert ("jac_user_kernel" if rule is None else "jac_user_chain_kernel") in RD.last_kernel_name(), RD.last_kernel_name()
    orule = O.CONTINUOUS if rule is None else {RD.Euler: O.EULER, RD.RK2: O.RK2, RD.RK3: O.RK3, RD.RK4: O.RK4}[rule]
    Jo, Yo = O.cartpole().jacobian(orule, O.USER, Zn, t, dt)
    np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
    Ja = RD.alloc_jacobians(fun, N, Z.data)
    RD.jacobian_(sig, RD.ForwardAD(), fun, Ja, None, Z)
    np.testing.assert_allclose(host(J), host(Ja), rtol=0, atol=1e-10)
    # several trajectories in one launch take the same body
    Zs = [RD.PackedTrajectory(4, 1, Zn[:300], t[:300], dt[:300]).to("cuda"), RD.PackedTrajectory(4, 1, Zn[300:], t[300:], dt[300:]).to("cuda")]
    Js = [RD.alloc_jacobians(fun, 300, Z.data), RD.alloc_jacobians(fun, N - 300, Z.data)]
    RD.jacobian_multi_(sig, RD.UserDefined(), fun, Js, None, Zs)
    np.testing.assert_array_equal(np.concatenate([host(Js[0]), host(Js[1])]), host(J))


# ------------------------------------------------------------------------------------------------------------------------------------------
# the round's committed fixtures (tests/golden/make_golden_round5.py; with Julia: ref_*.npz from make_golden_reference.jl take their place key by key)
# ------------------------------------------------------------------------------------------------------------------------------------------
def test_discrete_model_and_user_jacobian_fixtures():
    sig = RD.StaticReturn()
    d = H.load("discrete_di")
    N = d["Z"].shape[0]
    di = RD.CustomDiscreteModel(4, 2, DI_STEP, jacobian=DI_JAC)
    Z = RD.PackedTrajectory(4, 2, d["Z"], d["t"], d["dt"]).to("cuda")
    J = RD.alloc_jacobians(di, N, Z.data); Y = torch.empty((N, 4), dtype=torch.float64, device="cuda")
    for diff, key in ((RD.UserDefined(), "J_user"), (RD.ForwardAD(), "J_ad")):
        RD.jacobian_(sig, diff, di, J, Y, Z)
        np.testing.assert_allclose(host(J), d[key], rtol=0, atol=1e-15)
        np.testing.assert_allclose(host(Y), d["xn"], rtol=0, atol=1e-15)
    J1 = RD.alloc_jacobians(di, N - 1, Z.data); J2 = RD.alloc_jacobians(di, N - 1, Z.data)
    RD.dynamics_error_jacobian_(sig, RD.UserDefined(), di, J2, J1, None, None, Z)
    np.testing.assert_allclose(host(J1), d["J1"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(host(J2), d["J2"])
    np.testing.assert_allclose(host(RD.dynamics_error(di, Z)), d["e"], rtol=0, atol=1e-15)
    # the cartpole's hand-written Jacobian through UserDefined(): directly, and through every explicit rule's chain rule
    d = H.load("userjac_cartpole")
    N = d["Z"].shape[0]
    cart = RD.CustomModel(4, 1, H.CARTPOLE_SRC, params=list(d["params"]), jacobian=CART_JAC)
    Z = RD.PackedTrajectory(4, 1, d["Z"], d["t"], d["dt"]).to("cuda")
    for rule, key in ((None, "Jc_user"), (RD.Euler, "Jd_user_euler"), (RD.RK2, "Jd_user_rk2"), (RD.RK3, "Jd_user_rk3"), (RD.RK4, "Jd_user_rk4")):
        fun = cart if rule is None else RD.DiscretizedDynamics(cart, rule)
        J = RD.alloc_jacobians(fun, N, Z.data)
        RD.jacobian_(sig, RD.UserDefined(), fun, J, None, Z)
        np.testing.assert_allclose(host(J), d[key], rtol=0, atol=TOL * max(1.0, np.abs(d[key]).max()))


def test_nan_filled_knot_point_fixture():
    """A knot point as Traj(n, m, dt, N) creates it (NaN-filled, src/trajectories.jl:83) and dt = Inf on the last knot (:104-108): every other knot point's
    x_next and Jacobian are those of the clean batch (to the fixture's 1e-10), the poisoned one's are NaN."""
    d = H.load("nan_satellite")
    N = d["Z"].shape[0]
    sat = RD.Satellite(mass=float(d["mass"]), J=d["J"])
    dm = RD.DiscretizedDynamics(sat, RD.RK4)
    Z = RD.PackedTrajectory(13, 6, d["Z"], d["t"], d["dt"]).to("cuda")
    J = RD.alloc_jacobians(dm, N, Z.data); Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
    bad = np.concatenate([d["poisoned"], d["inf_dt"]])
    good = np.setdiff1d(np.arange(N), bad)
    np.testing.assert_allclose(host(J)[good], d["Jd_rk4"][good], rtol=0, atol=TOL)
    np.testing.assert_allclose(host(Y)[good], d["xn_rk4"][good], rtol=0, atol=TOL)
    p = int(d["poisoned"][0])
    assert np.isnan(host(Y)[p]).all() and np.isnan(host(J)[p]).any()


# ------------------------------------------------------------------------------------------------------------------------------------------
# fallback kernels (profiles/r05/kernel_reach.txt: templates the suite never launched): outputs that are only sizeof(T)-aligned, LieStates too wide for the tile kernels
# ------------------------------------------------------------------------------------------------------------------------------------------
def _misaligned(shape_cmajor, dtype=None):
    """a dense (N, c, r) stack whose first element sits 8 bytes behind a 16-byte boundary, as the (N, r, c) view the host functions take"""
    dtype = dtype or torch.float64
    n_el = int(np.prod(shape_cmajor))
    buf = torch.empty(n_el + 2, dtype=dtype, device="cuda")
    off = 1 if buf.data_ptr() % 16 == 0 else 0
    flat = buf[off:off + n_el]
    assert flat.data_ptr() % 16 == 8 or dtype != torch.float64
    return flat.view(*shape_cmajor).transpose(1, 2)


def test_outputs_that_are_not_16_byte_aligned_take_the_plain_tile_kernels():
    """The split-role kernels copy a tile as 16-byte units: a Jacobian stack that starts 8 bytes off such a boundary (a view into a bigger buffer) goes
    to the kernels whose waves compute and then copy (jac_ad_unrolled_kernel for models with a column plan, jac_ad_loop_kernel for the others), the
    error-state Jacobian to errstate_clear_kernel -- same numbers."""
    rng = np.random.default_rng(29)
    N = 20000                                              # (past the column-split kernel of small batches: 8 192 knot points on a whole MI355X)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    sat = RD.Satellite()
    dm = RD.DiscretizedDynamics(sat, RD.RK4)
    Zn = H.rand_rigid(rng, N, 6); t = np.arange(N) * 0.1; dt = np.full(N, 0.1)
    Z = RD.PackedTrajectory(13, 6, Zn, t, dt).to("cuda")
    Jo, Yo = O.satellite().jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
    J = _misaligned((N, 19, 13))
    RD.jacobian_(sig, AD, dm, J, None, Z)
    assert "jac_ad_unrolled_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL)
    # a model without a compile-time column plan and more than ten columns: RigidBody{MRP}
    mrp = RD.Satellite(R=RD.MRP)
    dmm = RD.DiscretizedDynamics(mrp, RD.RK4)
    Zm = np.concatenate([rng.random((N, 3)), 0.3 * rng.random((N, 3)), rng.random((N, 12))], axis=1)
    ZM = RD.PackedTrajectory(12, 6, Zm, t, dt).to("cuda")
    Ja = RD.alloc_jacobians(dmm, N, ZM.data)
    RD.jacobian_(sig, AD, dmm, Ja, None, ZM)
    aligned_name = RD.last_kernel_name()
    Jm = _misaligned((N, 18, 12))
    RD.jacobian_(sig, AD, dmm, Jm, None, ZM)
    assert "jac_ad_loop_kernel" in RD.last_kernel_name() and RD.last_kernel_name() != aligned_name, (RD.last_kernel_name(), aligned_name)
    np.testing.assert_allclose(host(Jm), host(Ja), rtol=0, atol=1e-13)
    # errstate_jacobian!, clear = True, G 8 bytes off
    quad = RD.Quadrotor()
    X = torch.as_tensor(H.rand_rigid(rng, N, 4)[:, :13]).cuda()
    Ga = RD.alloc_errstate_jacobians(quad, N, X)
    RD.errstate_jacobian_(quad, Ga, X, clear=True)
    G = _misaligned((N, 12, 13))
    G.fill_(7.0)
    RD.errstate_jacobian_(quad, G, X, clear=True)
    assert "errstate_clear_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    assert torch.equal(G, Ga)



def test_two_model_call_in_overlap_mode_repeats_without_fork_and_join():
    """rdhip_rollout_jacobian_multi on a context that owns its stream, launch overlap on (rdhip.h): the second model runs on the internal stream, a call
    that repeats the previous one argument for argument neither forks nor joins, and the next operation of any other kind -- a synchronise, a Jacobian
    call that reads what the pair produced -- is ordered behind all of it.  Results are those of the same call on a plain context."""
    from rdhip import _lib as L
    rng = np.random.default_rng(47)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    sat, cart = RD.Satellite(), RD.Cartpole()
    dsat, dcart = RD.DiscretizedDynamics(sat, RD.RK4), RD.DiscretizedDynamics(cart, RD.RK4)
    M, K = 1024, 96
    N = M * K
    t = np.tile(np.arange(K) * 0.01, (M, 1)); dt = np.full((M, K), 0.01)
    dsatd = (0.3 * H.rand_rigid(rng, N, 6) + np.array([0] * 3 + [0.7, 0, 0, 0] + [0] * 12)).reshape(M, K, 19)
    dcartd = 0.2 * rng.random((N, 5)).reshape(M, K, 5)
    f32 = lambda a: a.astype(np.float32)      # noqa: E731

    def build(ctx):
        st_s, st_c = _tm_stack(dsatd, t, dt, f32), _tm_stack(dcartd, t, dt, f32)
        Js, Jc = RD.alloc_jacobians(dsat, N, st_s["data"]), RD.alloc_jacobians(dcart, N, st_c["data"])
        return RD.PreparedRolloutJacobianMulti(sig, AD, [dcart, dsat], [Jc, Js], [st_c, st_s], None, time_major=True, ctx=ctx), st_s, st_c, Js, Jc
    plain, ps, pc, pJs, pJc = build(None)
    plain(); torch.cuda.synchronize()
    ctx = L.Context(0)
    ctx.set_overlap(True)
    call, st_s, st_c, Js, Jc = build(ctx)
    for rep in range(5):                                    # call 0 forks, calls 1 .. 4 are repetitions: nothing but kernels on either stream
        call()
    ctx.synchronize()
    for a, b in ((st_s["data"], ps["data"]), (st_c["data"], pc["data"]), (Js, pJs), (Jc, pJc)):
        assert torch.equal(a, b)
    # an operation of another kind right behind a repetition: a Jacobian call that overwrites the cartpole's J must come out on top
    other = RD.PackedTrajectory(4, 1, 0.2 * rng.random((N, 5)).astype(np.float32), np.arange(N) * 0.01, np.full(N, 0.01)).to("cuda")
    over = RD.prepare_jacobian(sig, AD, dcart, Jc, None, other, ctx=ctx)
    ref = RD.alloc_jacobians(dcart, N, st_c["data"])
    RD.jacobian_(sig, AD, dcart, ref, None, other); torch.cuda.synchronize()
    for rep in range(4):
        call(); call(); over()
        ctx.synchronize()
        assert torch.equal(Jc, ref), f"rep {rep}: the Jacobian call was overtaken by the pair's internal stream"
        assert torch.equal(Js, pJs)
    # ... and the pair right behind it starts from a fork again
    call(); ctx.synchronize()
    assert torch.equal(Jc, pJc) and torch.equal(Js, pJs)
    ctx.close()
