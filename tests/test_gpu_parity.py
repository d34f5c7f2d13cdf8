"""GPU parity tests: the HIP path (through the C ABI, via the rdhip host mirror) against the CPU oracle and the
committed golden vectors.  Tolerances: fp64 max |GPU - oracle| < 1e-10 (BASELINE.json north_star); fp32 is compared
with the fp64 oracle at rel 1e-4; integer queries are bit-exact (tests/test_host_abi.py)."""
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

RULES = {"euler": RD.Euler, "rk2": RD.RK2, "rk3": RD.RK3, "rk4": RD.RK4}
ORULE = {"euler": O.EULER, "rk2": O.RK2, "rk3": O.RK3, "rk4": O.RK4}


def rd_model(name, d):
    """Build the rdhip model that corresponds to a dyn_* fixture."""
    if name.startswith("cartpole"):
        return RD.Cartpole(*d["params"])
    if name.startswith("di"):
        return RD.DoubleIntegrator(int(d["params"][0]))
    if name.startswith("linear"):
        return RD.LinearModel(d["A"], d["B"])
    R = {0: RD.QuatRotation, 1: RD.MRP, 2: RD.RodriguesParam}[int(d["rot"]) if "rot" in d else 0]
    if name.startswith(("satellite_", "body_", "satfull_")):
        return RD.Satellite(float(d["mass"]), d["J"], R, bodyframe=bool(d["frame"]))
    if name.startswith("quadrotor"):
        return RD.Quadrotor(R, mass=float(d["mass"]), J=d["J"], gravity=d["gravity"], motor_dist=float(d["L"]), kf=float(d["kf"]), km=float(d["km"]),
                            bodyframe=bool(d["frame"]))
    if name.startswith("satpair"):
        return RD.SatellitePair(d["J1"], d["J2"])
    if name.startswith("liesat"):
        return RD.LieSatellite(d["Jd"])
    raise KeyError(name)


def traj(model, Z, t, dt, device=None, dtype=np.float64):
    n, m = RD.state_dim(model), RD.control_dim(model)
    T = RD.PackedTrajectory(n, m, np.ascontiguousarray(Z, dtype=dtype), t, dt)
    return T.to(device) if device else T


def host(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)


# ------------------------------------------------------------------------------------------------ golden fixtures
@pytest.mark.parametrize("name", H.dyn_cases())
@pytest.mark.parametrize("where", ["host", "cuda"])
def test_golden_dynamics_and_jacobians(name, where):
    d = H.load("dyn_" + name)
    model = rd_model(name, d)
    Z = traj(model, d["Z"], d["t"], d["dt"], "cuda" if where == "cuda" else None)
    N = len(Z)
    np.testing.assert_allclose(host(RD.dynamics(model, Z)), d["xdot"], rtol=0, atol=TOL)
    J = RD.alloc_jacobians(model, N, Z.data)
    Y = RD.evaluate(model, Z) * 0
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), model, J, Y, Z)
    np.testing.assert_allclose(host(J), d["Jc"], rtol=0, atol=TOL)
    np.testing.assert_allclose(host(Y), d["xdot"], rtol=0, atol=TOL)
    for rn, Q in RULES.items():
        dm = RD.DiscretizedDynamics(model, Q)
        np.testing.assert_allclose(host(RD.discrete_dynamics(dm, Z)), d["xn_" + rn], rtol=0, atol=TOL)
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
        np.testing.assert_allclose(host(J), d["Jd_" + rn], rtol=0, atol=TOL)
        np.testing.assert_allclose(host(Y), d["xn_" + rn], rtol=0, atol=TOL)
        # FiniteDifference: reference tolerance atol = 1e-6 (test/integration_tests.jl:14)
        RD.jacobian_(RD.InPlace(), RD.FiniteDifference(), dm, J, None, Z)
        scale = max(1.0, np.abs(d["Jd_" + rn]).max())
        np.testing.assert_allclose(host(J), d["Jd_" + rn], rtol=0, atol=1e-6 * scale)
        # UserDefined: the integrator's chain rule over continuous stage Jacobians == AD through the step to 1e-10
        # (test/integration_tests.jl:7-18); models without a user Jacobian raise NotImplementedError (src/functionbase.jl:257)
        if name.startswith(("satpair", "liesat")) or H.three_parameter(name):
            with pytest.raises(NotImplementedError):
                RD.jacobian_(RD.StaticReturn(), RD.UserDefined(), dm, J, Y, Z)
        else:
            J.fill(7.0) if isinstance(J, np.ndarray) else J.fill_(7.0)
            RD.jacobian_(RD.StaticReturn(), RD.UserDefined(), dm, J, Y, Z)
            np.testing.assert_allclose(host(J), d["Jd_" + rn], rtol=0, atol=TOL * scale)
            np.testing.assert_allclose(host(Y), d["xn_" + rn], rtol=0, atol=TOL)
    if not name.startswith(("satpair", "liesat")) and not H.three_parameter(name):
        RD.jacobian_(RD.StaticReturn(), RD.UserDefined(), model, J, Y, Z)
        np.testing.assert_allclose(host(J), d["Jc"], rtol=0, atol=TOL * max(1.0, np.abs(d["Jc"]).max()))


@pytest.mark.parametrize("tag,mk", [("rigid", RD.Satellite), ("satpair", RD.SatellitePair), ("liesat", RD.LieSatellite)])
@pytest.mark.parametrize("where", ["host", "cuda"])
def test_golden_errstate(tag, mk, where):
    d = H.load("err_" + tag)
    model = mk()
    X, Xb, X0 = d["X"], d["Xbar"], d["X0"]
    if where == "cuda":
        X, Xb, X0 = (torch.as_tensor(a).cuda() for a in (X, Xb, X0))
    N = X.shape[0]
    G = RD.alloc_errstate_jacobians(model, N, X)
    RD.errstate_jacobian_(model, G, X)                      # G pre-zeroed, like the reference's callers do
    np.testing.assert_allclose(host(G), d["G"], rtol=0, atol=TOL)
    G2 = RD.alloc_errstate_jacobians(model, N, X)
    G2 += 7.0
    RD.errstate_jacobian_(model, G2, X, clear=True)
    np.testing.assert_allclose(host(G2), d["G"], rtol=0, atol=TOL)
    np.testing.assert_allclose(host(RD.state_diff(model, X, X0)), d["dX"], rtol=TOL, atol=TOL)      # (relative where the Cayley error v / w is large: rdhip.h)
    e = RD.errstate_dim(model)
    base = torch.zeros((N, e, e), dtype=torch.float64, device="cuda") if where == "cuda" else np.zeros((N, e, e))
    dG = base.transpose(1, 2) if where == "cuda" else base.transpose(0, 2, 1)
    RD.nabla_errstate_jacobian_(model, dG, X, Xb)
    np.testing.assert_allclose(host(dG), d["dG"], rtol=0, atol=TOL)


@pytest.mark.parametrize("rtag", ["mrp", "rp"])
@pytest.mark.parametrize("where", ["host", "cuda"])
def test_golden_errstate_three_parameter_rotations(rtag, where):
    """errstate_jacobian! / state_diff of LieState{MRP,P} and LieState{RodriguesParam,P} (src/liestate.jl:262-298, :210-260): the rigid body's (3, 6)
    and test/liestate.jl:56-68's (3, 2, 3), against blocks differentiated from the composition map params(R * R(delta)) (make_golden_rot3.py) and
    against the oracle's closed forms; G is not cleared unless asked; the second-order map against the Hessian of the same composition map."""
    R = RD.MRP if rtag == "mrp" else RD.RodriguesParam
    for tag, what, om in (("rigid", RD.Body(R), O.satellite(2.0, np.diag([2.0, 3.0, 1.0]), rot=O.ROT_MRP if rtag == "mrp" else O.ROT_RP)), ("p323", RD.LieState(R, (3, 2, 3)), None)):
        d = H.load(f"err_{tag}_{rtag}")
        X, X0 = d["X"], d["X0"]
        if where == "cuda":
            X, X0 = torch.as_tensor(X).cuda(), torch.as_tensor(X0).cuda()
        N, n = X.shape
        e = RD.errstate_dim(what)
        assert e == n and d["G"].shape == (N, n, e)
        mk = lambda fill: ((torch.full((N, e, n), fill, dtype=torch.float64, device="cuda") if where == "cuda" else np.full((N, e, n), fill)).transpose(1, 2) if where == "cuda"
                           else np.full((N, e, n), fill).transpose(0, 2, 1))
        G = mk(0.0)
        RD.errstate_jacobian_(what, G, X)
        np.testing.assert_allclose(host(G), d["G"], rtol=0, atol=TOL)
        G2 = mk(7.0)
        RD.errstate_jacobian_(what, G2, X, clear=True)
        np.testing.assert_allclose(host(G2), d["G"], rtol=0, atol=TOL)
        G3 = mk(7.0)
        RD.errstate_jacobian_(what, G3, X)                     # pattern only: everything outside the ones and the 3 x 3 blocks keeps its 7
        g3 = host(G3)
        pattern = d["G"] != 0
        np.testing.assert_allclose(g3[pattern], d["G"][pattern], rtol=0, atol=TOL)
        assert np.all(g3[~pattern & (np.abs(d["G"]) == 0)][:5] == 7.0)
        np.testing.assert_allclose(host(RD.state_diff(what, X, X0)), d["dX"], rtol=TOL, atol=TOL)
        if om is not None:
            np.testing.assert_allclose(host(G), om.errstate_jacobian(d["X"]), rtol=0, atol=TOL)
        # nabla-errstate_jacobian! (src/liestate.jl:300-320): grad^2-differential blocks against the Hessian of delta -> xbar . (x (+) delta) (nab_*.npz), the
        # oracle's closed forms, and "only the blocks are written" without clear
        dn = H.load(f"nab_{tag}_{rtag}")
        assert np.array_equal(dn["X"], d["X"])
        Xb = torch.as_tensor(dn["Xbar"]).cuda() if where == "cuda" else dn["Xbar"]
        mkH = lambda fill: (torch.full((N, e, e), fill, dtype=torch.float64, device="cuda").transpose(1, 2) if where == "cuda" else np.full((N, e, e), fill).transpose(0, 2, 1))     # noqa: E731
        Hc = mkH(7.0)
        RD.nabla_errstate_jacobian_(what, Hc, X, Xb, clear=True)
        np.testing.assert_allclose(host(Hc), dn["dG"], rtol=0, atol=TOL * max(1.0, np.abs(dn["dG"]).max()))
        Hp = mkH(7.0)
        RD.nabla_errstate_jacobian_(what, Hp, X, Xb)               # blocks only
        hp = host(Hp)
        blk = np.zeros((e, e), dtype=bool)
        c0 = 0
        for i, pv in enumerate(tuple(int(v) for v in dn["P"])):
            c0 += pv
            if i < len(dn["P"]) - 1:
                blk[c0:c0 + 3, c0:c0 + 3] = True; c0 += 3
        np.testing.assert_allclose(hp[:, blk], dn["dG"][:, blk], rtol=0, atol=TOL * max(1.0, np.abs(dn["dG"]).max()))
        assert np.all(hp[:, ~blk] == 7.0)
        if om is not None:
            np.testing.assert_allclose(host(Hc), om.nabla_errstate_jacobian(dn["X"], dn["Xbar"]), rtol=0, atol=TOL * max(1.0, np.abs(dn["dG"]).max()))


def test_errstate_does_not_clear_by_default():
    # src/liestate.jl:293-297: only ones and rotation blocks are written
    model = RD.Body()
    rng = np.random.default_rng(0)
    X = H.rand_rigid(rng, 70, 6)[:, :13]
    G = RD.alloc_errstate_jacobians(model, 70, X)
    G += 7.0
    RD.errstate_jacobian_(model, G, X)
    ref = O.body().errstate_jacobian(X, clear=False, G0=np.full((70, 13, 12), 7.0))
    np.testing.assert_allclose(G, ref, rtol=0, atol=TOL)
    assert G[3, 0, 5] == 7.0 and G[3, 0, 0] == 1.0
    # Euclidean model: always G = I
    C = RD.Cartpole()
    Gc = RD.alloc_errstate_jacobians(C, 5, X)
    Gc += 3.0
    RD.errstate_jacobian_(C, Gc, rng.random((5, 4)))
    np.testing.assert_array_equal(Gc, np.broadcast_to(np.eye(4), (5, 4, 4)))


# ------------------------------------------------------------------------------------------------ seeded random vs oracle
CASES = [
    ("cartpole", lambda: RD.Cartpole(), lambda: O.cartpole(), lambda rng, N: rng.random((N, 5))),
    ("satellite", lambda: RD.Satellite(), lambda: O.satellite(), lambda rng, N: H.rand_rigid(rng, N, 6)),
    ("body_bodyframe", lambda: RD.Body(bodyframe=True), lambda: O.body(O.FRAME_BODY), lambda rng, N: H.rand_rigid(rng, N, 6)),
    ("quadrotor", lambda: RD.Quadrotor(), lambda: O.quadrotor(), lambda rng, N: H.rand_rigid(rng, N, 4, 0.6, 1.9)),
    ("quadrotor_body", lambda: RD.Quadrotor(bodyframe=True), lambda: O.quadrotor(O.FRAME_BODY), lambda rng, N: H.rand_rigid(rng, N, 4, 0.6, 1.9)),
]


@pytest.mark.parametrize("name,mk,omk,gen", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("N", [1, 63, 64, 65, 1000])
def test_random_vs_oracle_ragged_sizes(name, mk, omk, gen, N):
    rng = np.random.default_rng(N)
    model, om = mk(), omk()
    Zn = gen(rng, N)
    t = np.arange(N) * 0.1
    dt = np.full(N, 0.1)
    if N > 2:
        dt[-1] = 0.0          # terminal knot point: control reads as zeros (src/knotpoint.jl:67)
    Z = traj(model, Zn, t, dt, "cuda")
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    J = RD.alloc_jacobians(dm, N, Z.data)
    Y = torch.empty((N, RD.state_dim(model)), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
    Jo, Yo = om.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
    np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(RD.discrete_dynamics(RD.DiscretizedDynamics(model, RD.RK3), Z)), om.discrete_dynamics(O.RK3, Zn, t, dt), rtol=0, atol=TOL)
    # UserDefined chain rule (src/integration.jl:302-337) against the oracle's literal matrix form of the same formulas
    J2 = RD.alloc_jacobians(dm, N, Z.data)
    RD.jacobian_(RD.StaticReturn(), RD.UserDefined(), dm, J2, Y, Z)
    Ju, Yu = om.jacobian(O.RK4, O.USER, Zn, t, dt)
    np.testing.assert_allclose(host(J2), Ju, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(Y), Yu, rtol=0, atol=TOL)
    if N > 2:
        n = RD.state_dim(model)
        assert np.all(host(J)[-1, :, n:] == 0.0)
        np.testing.assert_array_equal(host(J)[-1, :, :n], np.eye(n))


def test_unaligned_output_slabs_take_the_fallback_kernel():
    """The split-role kernel copies tiles as 16-byte units and needs 16-byte aligned J / Y; an 8-byte aligned slab (a view into
    a larger buffer, as a Julia `view` would produce) must still give identical results through the element-wise kernel."""
    rng = np.random.default_rng(21)
    N = 333
    model, om = RD.Satellite(), O.satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    Zn = H.rand_rigid(rng, N, 6)
    t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
    Z = traj(model, Zn, t, dt, "cuda")
    Jo, Yo = om.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
    for diff in (RD.ForwardAD(), RD.UserDefined(), RD.FiniteDifference()):
        bufJ = torch.zeros(N * 247 + 1, dtype=torch.float64, device="cuda")
        bufY = torch.zeros(N * 13 + 1, dtype=torch.float64, device="cuda")
        J = bufJ[1:].view(N, 19, 13).transpose(1, 2)          # data_ptr is 8 mod 16
        Y = bufY[1:].view(N, 13)
        assert J.data_ptr() % 16 == 8 and Y.data_ptr() % 16 == 8
        RD.jacobian_(RD.StaticReturn(), diff, dm, J, Y, Z)
        np.testing.assert_allclose(host(J), Jo, rtol=0, atol=1e-6 if isinstance(diff, RD.FiniteDifference) else TOL)
        np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
        assert float(bufJ[0]) == 0.0 and float(bufY[0]) == 0.0   # nothing written in front of the slab


def test_prepared_jacobian_matches_the_checked_call():
    rng = np.random.default_rng(22)
    N = 200
    model = RD.Satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    Z = traj(model, H.rand_rigid(rng, N, 6), np.zeros(N), np.full(N, 0.1), "cuda")
    J1, J2 = RD.alloc_jacobians(dm, N, Z.data), RD.alloc_jacobians(dm, N, Z.data)
    Y1, Y2 = torch.empty((N, 13), dtype=torch.float64, device="cuda"), torch.empty((N, 13), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J1, Y1, Z)
    call = RD.prepare_jacobian(RD.StaticReturn(), RD.ForwardAD(), dm, J2, Y2, Z)
    call()
    assert torch.equal(J1, J2) and torch.equal(Y1, Y2)
    Z.data.mul_(0.5)                       # contents may change between calls
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J1, Y1, Z)
    call()
    assert torch.equal(J1, J2) and torch.equal(Y1, Y2)
    with pytest.raises(ValueError):
        RD.prepare_jacobian(RD.StaticReturn(), RD.ForwardAD(), dm, J2[:, :, :5], Y2, Z)
    # implicit rules: the diff method is the implicit function theorem, for the prepared call as for the checked one
    dmi = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
    ift = RD.ImplicitFunctionTheorem(RD.ForwardAD())
    RD.jacobian_(RD.StaticReturn(), ift, dmi, J1, Y1, Z)
    RD.prepare_jacobian(RD.StaticReturn(), ift, dmi, J2, Y2, Z)()
    assert torch.equal(J1, J2) and torch.equal(Y1, Y2)
    with pytest.raises(ValueError):
        RD.prepare_jacobian(RD.StaticReturn(), RD.ForwardAD(), dmi, J2, Y2, Z)


def test_degenerate_batch_sizes_of_the_pair_operators():
    """N = 0, 1 (no consecutive pair): dynamics_error, dynamics_error_jacobian!, error expansion are no-ops with empty outputs;
    N = 2 is the smallest real case."""
    model, om = RD.Satellite(), O.satellite()
    rng = np.random.default_rng(23)
    for Q in (RD.RK4, RD.ImplicitMidpoint):
        dm = RD.DiscretizedDynamics(model, Q)
        for N in (0, 1, 2):
            Zn = H.rand_rigid(rng, N, 6) if N else np.zeros((0, 19))
            Z = traj(model, Zn, np.zeros(N), np.full(N, 0.1), "cuda")
            E = RD.dynamics_error(dm, Z)
            assert tuple(E.shape) == (max(N - 1, 0), 13)
            J2, J1 = RD.alloc_jacobians(dm, max(N - 1, 0), Z.data), RD.alloc_jacobians(dm, max(N - 1, 0), Z.data)
            RD.dynamics_error_jacobian_(RD.StaticReturn(), RD.default_diffmethod(dm), dm, J2, J1, None, None, Z)
            if Q is RD.RK4:
                Eb = RD.alloc_error_expansions(dm, N, Z.data)
                RD.error_expansion_(dm, Eb, None, Z)
                if N == 2:
                    np.testing.assert_allclose(host(Eb), om.error_expansion(O.RK4, Zn, np.zeros(N), np.full(N, 0.1))[0], rtol=0, atol=TOL)
            if N == 2 and Q is RD.RK4:
                np.testing.assert_allclose(host(E), om.dynamics_error(O.RK4, Zn, np.zeros(N), np.full(N, 0.1)), rtol=0, atol=TOL)
    with pytest.raises(ValueError):
        RD.error_expansion_(RD.DiscretizedDynamics(model, RD.ImplicitMidpoint), RD.alloc_error_expansions(model, 3, None), None,
                            traj(model, H.rand_rigid(rng, 3, 6), np.zeros(3), np.full(3, 0.1), None))


def test_two_streams_two_contexts_run_concurrently():
    """One context per HIP stream (rdhip borrows torch's current stream): interleaved launches on two streams give the same results
    as sequential ones, and from two host threads (one context each, as the header prescribes) as well."""
    import threading
    rng = np.random.default_rng(24)
    model, om = RD.Satellite(), O.satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    N = 3000
    Zs = [H.rand_rigid(rng, N, 6) for _ in range(2)]
    t, dt = np.zeros(N), np.full(N, 0.1)
    refs = [om.jacobian(O.RK4, O.FORWARD_AD, Z, t, dt)[0] for Z in Zs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    Zd = [traj(model, Z, t, dt, "cuda") for Z in Zs]
    Js = [RD.alloc_jacobians(dm, N, Zd[0].data) for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(5):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, Js[i], None, Zd[i])
    torch.cuda.synchronize()
    for i in (0, 1):
        np.testing.assert_allclose(host(Js[i]), refs[i], rtol=0, atol=TOL)
        Js[i].zero_()
    errs = []

    def worker(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(20):
                    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, Js[i], None, Zd[i])
            streams[i].synchronize()
        except Exception as e:      # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=worker, args=(i,)) for i in (0, 1)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errs
    for i in (0, 1):
        np.testing.assert_allclose(host(Js[i]), refs[i], rtol=0, atol=TOL)


def test_launch_overlap_keeps_dependent_calls_in_order():
    """rdhip_ctx_set_overlap: consecutive rdhip_jacobian calls of a context alternate between two streams.  Independent calls may overlap; calls
    whose arrays intersect (write after read, read after write, write after write) must still behave as if executed in order, other
    operations on the context and rdhip_ctx_fence / synchronize must see everything issued before them, and a borrowed stream is refused."""
    from rdhip import _lib as L
    rng = np.random.default_rng(5)
    model, om = RD.Satellite(), O.satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    N = 20000
    ctx = L.Context(0)
    ctx.set_overlap(True)
    with pytest.raises(NotImplementedError):
        L.context_for(0, torch.cuda.current_stream(0).cuda_stream).set_overlap(True)
    Zn = [H.rand_rigid(rng, N, 6) for _ in range(3)]
    t, dt = np.arange(N) * 0.05, np.full(N, 0.05)
    Zs = [traj(model, z, t, dt, "cuda") for z in Zn]
    refs = [om.jacobian(O.RK4, O.FORWARD_AD, z, t, dt) for z in Zn]
    torch.cuda.synchronize()
    # (1) independent calls, rotating outputs
    Js = [RD.alloc_jacobians(dm, N, Zs[0].data) for _ in range(3)]
    Ys = [torch.empty((N, 13), dtype=torch.float64, device="cuda") for _ in range(3)]
    calls = [RD.prepare_jacobian(sig, AD, dm, Js[i], Ys[i], Zs[i], ctx=ctx) for i in range(3)]
    for rep in range(4):
        for c in calls:
            c()
    ctx.synchronize()
    for i in range(3):
        np.testing.assert_allclose(host(Js[i]), refs[i][0], rtol=0, atol=TOL)
        np.testing.assert_allclose(host(Ys[i]), refs[i][1], rtol=0, atol=TOL)
    # (2) write after write on ONE output: the last call wins, every time
    J = RD.alloc_jacobians(dm, N, Zs[0].data)
    same = [RD.prepare_jacobian(sig, AD, dm, J, None, Zs[i], ctx=ctx) for i in range(3)]
    for rep in range(5):
        for c in same:
            c()
        ctx.synchronize()
        np.testing.assert_allclose(host(J), refs[2][0], rtol=0, atol=TOL)
    # (3) read after write: call B's time steps ARE memory call A writes (the head of A's f(z) output): B must see A's values
    Ya = torch.zeros((N, 13), dtype=torch.float64, device="cuda")
    JA, JB = RD.alloc_jacobians(dm, N, Zs[0].data), RD.alloc_jacobians(dm, N, Zs[0].data)
    A_call = RD.prepare_jacobian(sig, AD, dm, JA, Ya, Zs[1], ctx=ctx)
    dtB = Ya.view(-1)[:N]
    B_call = RD.prepare_jacobian(sig, AD, dm, JB, None, RD.PackedTrajectory(13, 6, Zs[0].data, Zs[0].t, dtB), ctx=ctx)
    dt_expected = refs[1][1].reshape(-1)[:N]
    JBo = om.jacobian(O.RK4, O.FORWARD_AD, Zn[0], t, dt_expected)[0]
    for rep in range(4):
        Ya.zero_()
        torch.cuda.synchronize()
        A_call(); B_call()
        ctx.synchronize()
        np.testing.assert_allclose(host(JB), JBo, rtol=0, atol=1e-9)
    # (4) write after read: the second call overwrites (with its Jacobians) the array the first one reads as its time steps
    Jc = RD.alloc_jacobians(dm, N, Zs[0].data)
    big = torch.zeros(N * 13 * 19, dtype=torch.float64, device="cuda")
    reader = RD.prepare_jacobian(sig, AD, dm, Jc, None, RD.PackedTrajectory(13, 6, Zs[2].data, Zs[2].t, big[:N]), ctx=ctx)
    writer = RD.prepare_jacobian(sig, AD, dm, big.view(N, 19, 13).transpose(1, 2), None, Zs[0], ctx=ctx)
    for rep in range(4):
        big[:N].copy_(Zs[2].dt)
        torch.cuda.synchronize()
        reader(); writer()                                   # the reader must have consumed its time steps before the writer overwrites them
        ctx.synchronize()
        np.testing.assert_allclose(host(Jc), refs[2][0], rtol=0, atol=TOL)
        np.testing.assert_allclose(host(big.view(N, 19, 13).transpose(1, 2)), refs[0][0], rtol=0, atol=TOL)
    ctx.set_overlap(False)
    ctx.close()


def test_device_calls_can_be_captured_in_a_hip_graph():
    """A solver iteration at small N is launch-bound; with device-resident data the library only enqueues kernels on the context's
    stream, so a sequence of calls can be captured once into a HIP graph (torch.cuda.graph on ROCm) and replayed.  The replay must give
    what the eager calls give, also after the inputs changed in place."""
    rng = np.random.default_rng(27)
    N = 101
    model, om = RD.Quadrotor(), O.quadrotor()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    Zn = H.rand_rigid(rng, N, 4)
    t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
    Z = traj(model, Zn, t, dt, "cuda")
    J = RD.alloc_jacobians(dm, N, Z.data)
    Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    G = RD.alloc_errstate_jacobians(model, N, Z.data)
    E = RD.alloc_error_expansions(dm, N, Z.data)
    s = torch.cuda.Stream()

    def iteration():
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
        RD.errstate_jacobian_(model, G, Z.data[:, :13], clear=True)
        RD.error_expansion_(dm, E, None, Z)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        iteration()                                   # warm-up: creates the context of this stream, sets the kernels' LDS attributes
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        iteration()
    for scale in (1.0, 0.5):
        Z.data[:, 7:].mul_(scale)                     # velocities and controls change in place between replays
        Zh = host(Z.data)
        J.zero_(); Y.zero_(); G.zero_(); E.zero_()
        g.replay()
        torch.cuda.synchronize()
        Jr, Yr = om.jacobian(O.RK4, O.FORWARD_AD, Zh, t, dt)
        np.testing.assert_allclose(host(J), Jr, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(Y), Yr, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(G), om.errstate_jacobian(Zh[:, :13]), rtol=0, atol=TOL)
        np.testing.assert_allclose(host(E), om.error_expansion(O.RK4, Zh, t, dt)[0], rtol=0, atol=TOL)


def test_c_abi_graph_capture_and_replay():
    """rdhip_graph_begin_capture / end_capture / launch on a context that owns its stream (what a Julia host without a HIP API uses):
    the captured sequence (RK4 Jacobians + x_next, discrete dynamics, error-state Jacobians) replays with one launch and follows
    in-place changes of the inputs; host-memory calls are refused while capturing."""
    from rdhip import _lib as L
    rng = np.random.default_rng(28)
    N = 300
    model, om = RD.Satellite(), O.satellite()
    Zn, t, dt = H.rand_rigid(rng, N, 6), np.arange(N) * 0.1, np.full(N, 0.1)
    Z = traj(model, Zn, t, dt, "cuda")
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    J = RD.alloc_jacobians(dm, N, Z.data)
    Y, X2 = torch.empty((N, 13), dtype=torch.float64, device="cuda"), torch.empty((N, 13), dtype=torch.float64, device="cuda")
    G = RD.alloc_errstate_jacobians(model, N, Z.data)
    ctx = L.Context(0)                                   # owns a non-blocking stream
    lib = L.lib()
    ptrs = (Z.data.data_ptr(), Z.t.data_ptr(), Z.dt.data_ptr())

    def sequence():
        L.check(lib.rdhip_jacobian(ctx.handle, model._handle, L.RK4, L.FORWARD_AD, L.F64, N, *ptrs, J.data_ptr(), Y.data_ptr(), L.DEVICE), ctx.handle)
        L.check(lib.rdhip_discrete_dynamics(ctx.handle, model._handle, L.RK4, L.F64, N, *ptrs, X2.data_ptr(), L.DEVICE), ctx.handle)
        L.check(lib.rdhip_errstate_jacobian(ctx.handle, model._handle, L.F64, N, Z.data.data_ptr(), 19, 1, G.data_ptr(), L.DEVICE), ctx.handle)
        # an implicit step: memset of the context's flag word + register kernel + gated LDS kernel, all capturable
        L.check(lib.rdhip_discrete_dynamics(ctx.handle, model._handle, L.IMPLICIT_MIDPOINT, L.F64, N, *ptrs, X3.data_ptr(), L.DEVICE), ctx.handle)
    X3 = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    sequence(); ctx.synchronize()                       # first use outside the capture
    ctx.begin_capture()
    sequence()
    with pytest.raises(ValueError):                     # a host-memory call cannot be captured
        RDh = np.empty((N, 13))
        L.check(lib.rdhip_discrete_dynamics(ctx.handle, model._handle, L.RK4, L.F64, N, Zn.ctypes.data, t.ctypes.data, dt.ctypes.data, RDh.ctypes.data, L.HOST), ctx.handle)
    graph = ctx.end_capture()
    for scale in (1.0, 0.25):
        Z.data[:, 7:].mul_(scale)
        J.zero_(); Y.zero_(); X2.zero_(); G.zero_(); X3.zero_()
        torch.cuda.synchronize()
        ctx.launch(graph); ctx.launch(graph)            # replays are idempotent
        ctx.synchronize()
        Zh = host(Z.data)
        Jr, Yr = om.jacobian(O.RK4, O.FORWARD_AD, Zh, t, dt)
        np.testing.assert_allclose(host(J), Jr, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(Y), Yr, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(X2), Yr, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(G), om.errstate_jacobian(Zh[:, :13]), rtol=0, atol=TOL)
        np.testing.assert_allclose(host(X3), om.implicit_midpoint(Zh, t, dt, want_jacobian=False)[0], rtol=0, atol=1e-9)
    with pytest.raises(ValueError):
        ctx.end_capture()                               # not capturing
    graph.close(); ctx.close()


def test_one_process_sharded_jacobian_and_peer_copy():
    """rdhip_jacobian_sharded: host arrays, contiguous blocks of knot points on several contexts of ONE process (here: three contexts on
    the only GPU of the test box, ragged blocks incl. an empty one for tiny N), concurrently; rdhip_memcpy_peer stacks device shards."""
    from rdhip import _lib as L
    import ctypes as C
    rng = np.random.default_rng(29)
    model, om = RD.Quadrotor(), O.quadrotor()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    for N in (1000, 2, 0):
        Zn, t, dt = H.rand_rigid(rng, N, 4) if N else np.zeros((0, 17)), np.arange(N) * 0.1, np.full(N, 0.1)
        Z = traj(model, Zn, t, dt, None)
        J = RD.alloc_jacobians(dm, N, Z.data)
        Y = np.empty((N, 13))
        RD.jacobian_sharded_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z, devices=[0, 0, 0])
        if N:
            Jr, Yr = om.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
            np.testing.assert_allclose(J, Jr, rtol=0, atol=TOL)
            np.testing.assert_allclose(Y, Yr, rtol=0, atol=TOL)
    lo, cnt = C.c_longlong(), C.c_longlong()
    spans = []
    for i in range(3):
        L.check(L.lib().rdhip_shard_range(1000, 3, i, C.byref(lo), C.byref(cnt)))
        spans.append((lo.value, lo.value + cnt.value))
    assert spans == [RD.parallel.shard_range(1000, i, 3) for i in range(3)] == [(0, 334), (334, 668), (668, 1000)]
    with pytest.raises(TypeError):                      # device arrays are the business of rdhip.parallel / one context per device
        RD.jacobian_sharded_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, traj(model, np.zeros((0, 17)), np.zeros(0), np.zeros(0), "cuda"), devices=[0])
    # peer copy (same device here): ordered after the producer stream, runs on the consumer stream
    a, b = L.Context(0), L.Context(0)
    src = torch.arange(4096, dtype=torch.float64, device="cuda")
    dst = torch.zeros(4096, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    L.check(L.lib().rdhip_memcpy_peer(b.handle, dst.data_ptr(), a.handle, src.data_ptr(), 4096 * 8), b.handle)
    b.synchronize()
    assert torch.equal(src, dst)
    a.close(); b.close()


def test_sincos_argument_reduction_over_the_whole_range():
    """scalar.hpp's fp64 sincos_ (Cody-Waite reduction, < 8e5) and its library fallback (beyond): the cartpole's dynamics and RK4 Jacobian at
    pole angles from 1e-300 to 1e15 must match the oracle (glibc sin / cos) -- a wrong reduction constant shows up at large angles."""
    rng = np.random.default_rng(33)
    mags = np.array([1e-300, 1e-8, 0.5, 3.0, 1e2, 12345.678, 7.9e5, 7.99999e5, 8.0e5, 8.00001e5, 1e6, 1e9, 1e15])
    th = np.concatenate([mags, -mags, rng.uniform(-8e5, 8e5, 3000), rng.uniform(-50, 50, 3000), np.arange(1, 2000) * (np.pi / 2)])
    N = th.size
    Zn = rng.random((N, 5)); Zn[:, 1] = th
    t, dt = np.zeros(N), np.full(N, 1e-3)
    model, om = RD.Cartpole(), O.cartpole()
    Z = traj(model, Zn, t, dt, "cuda")
    np.testing.assert_allclose(host(RD.dynamics(model, Z)), om.dynamics(Zn, t, dt), rtol=1e-13, atol=1e-12)
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    J = RD.alloc_jacobians(dm, N, Z.data)
    Y = torch.empty((N, 4), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
    Jo, Yo = om.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
    np.testing.assert_allclose(host(Y), Yo, rtol=1e-13, atol=1e-9)
    np.testing.assert_allclose(host(J), Jo, rtol=1e-12, atol=TOL)
    # fp32 has its own sincos_ (three-constant reduction below 1e5, library beyond): a few float ulps of the fp64 oracle at angles float can resolve
    th32 = np.concatenate([np.array([1e-30, 1e-4, 0.5, 3.0, 1e2, 12345.678, 9.9e4, 99999.9, 1.0e5, 100000.1, 1e6]), rng.uniform(-1e5, 1e5, 4000),
                           rng.uniform(-50, 50, 3000), np.arange(1, 2000) * (np.pi / 2)]).astype(np.float32)
    th32 = np.concatenate([th32, -th32])
    Z32 = rng.random((th32.size, 5)).astype(np.float32); Z32[:, 1] = th32
    t32, dt32 = np.zeros(th32.size), np.full(th32.size, 1e-3)
    got = host(RD.dynamics(model, traj(model, Z32, t32, dt32, "cuda", dtype=np.float32))).astype(np.float64)
    np.testing.assert_allclose(got, om.dynamics(Z32.astype(np.float64), t32, dt32), rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("where", ["host", "cuda"])
def test_state_diff_of_bare_liestates_quat_rp_mrp(where):
    """state_diff(s::LieState, x, x0) on raw rand(n) vectors, as test/liestate.jl:33-51 (QuatRotation, P = (3,2,3)) and :56-68
    (RodriguesParam): vector parts subtract, every rotation part is q (-) q0 = the Cayley (Rodrigues) parameters of q0^-1 q.
    Expected values from scipy's Rotation algebra (independent of the library and of the oracle)."""
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(33)
    N = 257

    def to_scipy(kind, p):
        if kind == "quat":
            return R.from_quat(np.roll(p / np.linalg.norm(p, axis=1, keepdims=True), -1, axis=1))     # scipy is scalar-last
        if kind == "rp":
            q = np.concatenate([np.ones((len(p), 1)), p], axis=1)
            return R.from_quat(np.roll(q / np.linalg.norm(q, axis=1, keepdims=True), -1, axis=1))
        return R.from_mrp(p)

    for Rtype, kind, rp in ((RD.QuatRotation, "quat", 4), (RD.RodriguesParam, "rp", 3), (RD.MRP, "mrp", 3)):
        P = (3, 2, 3)
        s = RD.LieState(Rtype, P)
        n = len(s)
        assert n == sum(P) + 2 * rp and RD.errstate_dim(s) == 14
        X, X0 = rng.random((N, n)), rng.random((N, n))
        exp = np.zeros((N, 14))
        r = c = 0
        for i, p in enumerate(P):
            exp[:, c:c + p] = X[:, r:r + p] - X0[:, r:r + p]; r += p; c += p
            if i < len(P) - 1:
                rel = (to_scipy(kind, X0[:, r:r + rp]).inv() * to_scipy(kind, X[:, r:r + rp])).as_quat()
                exp[:, c:c + 3] = rel[:, :3] / rel[:, 3:4]
                r += rp; c += 3
        a, b = (torch.as_tensor(X).cuda(), torch.as_tensor(X0).cuda()) if where == "cuda" else (X, X0)
        np.testing.assert_allclose(host(RD.state_diff(s, a, b)), exp, rtol=0, atol=1e-12)
        if kind != "quat":      # the second-order map of MRP / RodriguesParam states: symmetric 3 x 3 blocks, zero elsewhere
            Hn = np.zeros((N, 14, 14)).transpose(0, 2, 1)
            RD.nabla_errstate_jacobian_(s, Hn, X, X0, clear=True)
            Hn = np.asarray(Hn)
            np.testing.assert_allclose(Hn, Hn.transpose(0, 2, 1), rtol=0, atol=1e-14)
            mask = np.zeros((14, 14), dtype=bool); mask[3:6, 3:6] = True; mask[8:11, 8:11] = True
            assert np.all(Hn[:, ~mask] == 0) and np.abs(Hn[:, mask]).max() > 0


def test_empty_batch_is_a_noop():
    model = RD.Satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    Z = traj(model, np.zeros((0, 19)), np.zeros(0), np.zeros(0), "cuda")
    J = RD.alloc_jacobians(dm, 0, Z.data)
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, None, Z)
    assert RD.discrete_dynamics(dm, Z).shape == (0, 13)


def test_fp32_against_fp64_oracle():
    # no Julia fp32 twin exists (KnotPoint.t/dt are Float64): compare with the fp64 oracle, rel 1e-4
    rng = np.random.default_rng(5)
    for model, om, Zn in ((RD.Cartpole(), O.cartpole(), rng.random((500, 5))), (RD.Satellite(), O.satellite(), H.rand_rigid(rng, 500, 6))):
        N = 500
        t, dt = np.zeros(N), np.full(N, 0.1)
        Z = traj(model, Zn, t, dt, "cuda", dtype=np.float32)
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        J = RD.alloc_jacobians(dm, N, Z.data)
        assert J.dtype == torch.float32
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, None, Z)
        Jo, _ = om.jacobian(O.RK4, O.FORWARD_AD, Zn.astype(np.float32).astype(np.float64), t, dt)
        np.testing.assert_allclose(host(J), Jo, rtol=1e-4, atol=1e-4)


def test_full_size_satellite_properties_and_sampled_oracle():
    """BASELINE config: Satellite RK4, N = 65536.  Size-independent properties + oracle on the whole batch."""
    N = 65536
    rng = np.random.default_rng(3)
    Zn = H.rand_rigid(rng, N, 6)
    t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
    model = RD.Satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    Z = traj(model, Zn, t, dt, "cuda")
    J = RD.alloc_jacobians(dm, N, Z.data)
    Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
    Jh, Yh = host(J), host(Y)
    # structural facts of the world-frame satellite: d r'/d r = I, d r'/d v = h I, d v'/d v = I, rows q / w independent of r, v, f
    np.testing.assert_array_equal(Jh[:, 0:3, 0:3], np.broadcast_to(np.eye(3), (N, 3, 3)))
    np.testing.assert_allclose(Jh[:, 0:3, 7:10], np.broadcast_to(0.1 * np.eye(3), (N, 3, 3)), atol=1e-15)
    np.testing.assert_array_equal(Jh[:, 7:10, 7:10], np.broadcast_to(np.eye(3), (N, 3, 3)))
    assert not Jh[:, 3:7, 0:3].any() and not Jh[:, 3:7, 7:10].any() and not Jh[:, 3:7, 13:16].any()
    assert not Jh[:, 10:13, 0:10].any() and not Jh[:, 10:13, 13:16].any()
    # linearity in the control: x'(u) is affine in f and tau for fixed x (F = q*f, tau enters w linearly only through RK products)
    # -> check the directional derivative identity J*dz ~ (f(z+eps dz) - f(z-eps dz)) / 2eps on a slice
    sl = slice(0, 4096)
    dz = rng.standard_normal((4096, 19))
    eps = 1e-6
    om = O.satellite()
    fp = om.discrete_dynamics(O.RK4, Zn[sl] + eps * dz, t[sl], dt[sl])
    fm = om.discrete_dynamics(O.RK4, Zn[sl] - eps * dz, t[sl], dt[sl])
    np.testing.assert_allclose(np.einsum("kij,kj->ki", Jh[sl], dz), (fp - fm) / (2 * eps), atol=1e-7)
    # whole batch against the oracle
    Jo, Yo = om.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
    assert np.abs(Jh - Jo).max() < TOL and np.abs(Yh - Yo).max() < TOL


def test_full_size_quadrotor_errstate_properties_and_oracle():
    """BASELINE config 4: Quadrotor errstate_jacobian!, 262144 knot points (one rank's 32768-knot shard against the oracle, the whole
    batch through size-independent properties: G = blkdiag(I3, L(q)H, I6), (L(q)H)'(L(q)H) = I3, L(q)H ' q = 0)."""
    N = 262144
    rng = np.random.default_rng(4)
    Zn = H.rand_rigid(rng, N, 4, 0.6, 1.9)
    Zn[:, 3:7] *= rng.uniform(0.5, 2.0, (N, 1))             # un-normalised quaternions: QuatRotation normalises on construction
    model = RD.Quadrotor()
    Z = traj(model, Zn, np.arange(N) * 0.1, np.full(N, 0.1), "cuda")
    G = RD.alloc_errstate_jacobians(model, N, Z.data)
    G.fill_(7.0)
    RD.errstate_jacobian_(model, G, Z, clear=True)
    Gh = host(G)
    assert Gh.shape == (N, 13, 12)
    np.testing.assert_array_equal(Gh[:, 0:3, 0:3], np.broadcast_to(np.eye(3), (N, 3, 3)))
    np.testing.assert_array_equal(Gh[:, 7:13, 6:12], np.broadcast_to(np.eye(6), (N, 6, 6)))
    assert not Gh[:, 0:3, 3:].any() and not Gh[:, 3:7, 0:3].any() and not Gh[:, 3:7, 6:].any() and not Gh[:, 7:, 0:6].any()
    A = Gh[:, 3:7, 3:6]
    np.testing.assert_allclose(np.einsum("kia,kib->kab", A, A), np.broadcast_to(np.eye(3), (N, 3, 3)), atol=1e-14)
    qn = Zn[:, 3:7] / np.linalg.norm(Zn[:, 3:7], axis=1, keepdims=True)
    np.testing.assert_allclose(np.einsum("kia,ki->ka", A, qn), 0.0, atol=1e-15)
    sl = slice(5 * 32768, 6 * 32768)                          # rank 5's shard of the 8-way split
    np.testing.assert_allclose(Gh[sl], O.quadrotor().errstate_jacobian(Zn[sl, :13]), rtol=0, atol=TOL)
    # reference-exact mode (does not clear G): touches only the pattern entries
    G2 = RD.alloc_errstate_jacobians(model, 4096, Z.data)
    G2.fill_(7.0)
    Zs = traj(model, Zn[:4096], np.zeros(4096), np.full(4096, 0.1), "cuda")
    RD.errstate_jacobian_(model, G2, Zs)
    G2h = host(G2)
    pattern = Gh[:4096] != 0
    np.testing.assert_array_equal(G2h[pattern], Gh[:4096][pattern])
    assert np.all(G2h[~pattern] == 7.0)      # every entry outside the pattern is UNTOUCHED (a kernel that zero-filled them would fail here)
    # nabla errstate Jacobian: blkdiag(0, -(q.b) I3, 0)   (test/rigid_body_jacobians.jl:78-83)
    Xb = rng.random((4096, 13))
    dG = torch.full((4096, 12, 12), 7.0, dtype=torch.float64, device="cuda").transpose(1, 2)
    RD.nabla_errstate_jacobian_(model, dG, torch.as_tensor(Zn[:4096, :13]).cuda(), torch.as_tensor(Xb).cuda(), clear=True)
    dGh = host(dG)
    qb = -(qn[:4096] * Xb[:, 3:7]).sum(1)
    ref = np.zeros((4096, 12, 12))
    for j in range(3):
        ref[:, 3 + j, 3 + j] = qb
    np.testing.assert_allclose(dGh, ref, rtol=0, atol=1e-14)


def test_mixed_model_monte_carlo_fp32_shard():
    """BASELINE config 5 on one GPU's share: 1024 trajectories x 256 knots, even trajectories Cartpole / odd Satellite, fp32.
    Every trajectory is rolled out on the GPU (RK4) and the Jacobians are evaluated along the rollout; checked against the
    fp64 oracle on sampled trajectories (rel 1e-3: fp32 error accumulated over 255 steps) and through the linearisation
    identity x_{k+1} = f(z_k) (the rollout and the Jacobian kernel's f(z) output agree to fp32 rounding)."""
    rng = np.random.default_rng(5)
    M, K = 512, 256
    for model, om, nz, gen in ((RD.Cartpole(), O.cartpole(), 5, lambda n: 0.2 * rng.random((n, 5))),
                               (RD.Satellite(), O.satellite(), 19, lambda n: 0.3 * H.rand_rigid(rng, n, 6) + np.array([0] * 3 + [0.7, 0, 0, 0] + [0] * 12))):
        n = RD.state_dim(model)
        data = gen(M * K).reshape(M, K, nz).astype(np.float32)
        t = np.tile(np.arange(K) * 0.01, (M, 1))
        dt = np.full((M, K), 0.01)
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(t).cuda(), "dt": torch.as_tensor(dt).cuda()}
        RD.rollout_(RD.StaticReturn(), dm, stack)
        rolled = host(stack["data"])
        assert np.isfinite(rolled).all()
        sample = [0, 1, 255, 511]
        ref = om.rollout(O.RK4, data[sample].astype(np.float64), t[sample], dt[sample])
        np.testing.assert_allclose(rolled[sample][..., :n], ref[..., :n], rtol=2e-3, atol=2e-3)
        Z = RD.PackedTrajectory(n, nz - n, stack["data"].reshape(M * K, nz), stack["t"].reshape(-1), stack["dt"].reshape(-1))
        J = RD.alloc_jacobians(dm, M * K, Z.data)
        Y = torch.empty((M * K, n), dtype=torch.float32, device="cuda")
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
        Yh = host(Y).reshape(M, K, n)
        np.testing.assert_allclose(Yh[:, :-1], rolled[:, 1:, :n], rtol=1e-5, atol=1e-6)   # same formulas; FMA contraction differs by an ulp
        Jo, _ = om.jacobian(O.RK4, O.FORWARD_AD, rolled[3].astype(np.float64), t[3], dt[3])
        np.testing.assert_allclose(host(J).reshape(M, K, n, nz)[3], Jo, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("where", ["host", "cuda"])
def test_dynamics_error_and_jacobian(where):
    """dynamics_error / dynamics_error_jacobian! (src/discrete_dynamics.jl:134-201): e_k = f(z_k) - x_{k+1}, J1 = [A B], J2 = [-I 0];
    a rolled-out trajectory has zero dynamics error (test/discrete_dynamics_test.jl pattern)."""
    rng = np.random.default_rng(11)
    for model, om, Zn in ((RD.Cartpole(), O.cartpole(), rng.random((130, 5))), (RD.Satellite(), O.satellite(), H.rand_rigid(rng, 130, 6))):
        N = Zn.shape[0]
        n, m = RD.state_dim(model), RD.control_dim(model)
        t, dt = np.arange(N) * 0.05, np.full(N, 0.05)
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        Z = traj(model, Zn, t, dt, "cuda" if where == "cuda" else None)
        np.testing.assert_allclose(host(RD.dynamics_error(dm, Z)), om.dynamics_error(O.RK4, Zn, t, dt), rtol=0, atol=TOL)
        J2, J1 = RD.alloc_jacobians(dm, N - 1, Z.data), RD.alloc_jacobians(dm, N - 1, Z.data)
        y2 = RD.evaluate(dm, Z)[: N - 1] * 0 + 5
        for diff, tol in ((RD.ForwardAD(), TOL), (RD.UserDefined(), TOL), (RD.FiniteDifference(), 1e-5)):
            RD.dynamics_error_jacobian_(RD.StaticReturn(), diff, dm, J2, J1, y2, None, Z)
            J2o, J1o, eo = om.dynamics_error_jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
            np.testing.assert_array_equal(host(J2), J2o)
            np.testing.assert_allclose(host(J1), J1o, rtol=0, atol=tol * max(1.0, np.abs(J1o).max()))
            np.testing.assert_allclose(host(y2), eo, rtol=0, atol=TOL)
        # per-knot form reads like the reference
        z1, z2 = RD.KnotPoint(Zn[3, :n], Zn[3, n:], t[3], dt[3]), RD.KnotPoint(Zn[4, :n], Zn[4, n:], t[4], dt[4])
        np.testing.assert_allclose(RD.dynamics_error(dm, z2, z1), om.dynamics_error(O.RK4, Zn[3:5], t[3:5], dt[3:5])[0], rtol=0, atol=TOL)
        D2, D1 = RD.DynamicsJacobian(n, m), RD.DynamicsJacobian(n, m)
        RD.dynamics_error_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, D2, D1, np.zeros(n), np.zeros(n), z2, z1)
        np.testing.assert_allclose(np.asarray(D1), J1o[3], rtol=0, atol=TOL)
        np.testing.assert_array_equal(np.asarray(D2), J2o[3])
        # rollout => zero error
        stack = {"data": np.ascontiguousarray(Zn[None].copy()), "t": t[None].copy(), "dt": dt[None].copy()}
        if where == "cuda":
            stack = {k: torch.as_tensor(v).cuda() for k, v in stack.items()}
        RD.rollout_(RD.StaticReturn(), dm, stack)
        Zr = RD.PackedTrajectory(n, m, stack["data"][0], stack["t"][0], stack["dt"][0])
        assert np.abs(host(RD.dynamics_error(dm, Zr))).max() < 1e-13
    with pytest.raises(TypeError):
        RD.dynamics_error(RD.Cartpole(), Z)


@pytest.mark.parametrize("where", ["host", "cuda"])
def test_error_state_expansion_fused_into_the_jacobian_kernel(where):
    """E[k] = [G(x_{k+1})' A_k G(x_k) | G(x_{k+1})' B_k] (TrajectoryOptimization.jl error_expansion! built from this path's operators)
    against the oracle's explicit product of its own Jacobian and errstate Jacobians; Euclidean models return [A B]."""
    rng = np.random.default_rng(12)
    cases = [(RD.Satellite(), O.satellite(), lambda n: H.rand_rigid(rng, n, 6)),
             (RD.Body(bodyframe=True), O.body(O.FRAME_BODY), lambda n: H.rand_rigid(rng, n, 6)),
             (RD.Satellite(mass=1.3, J=np.array([[2.0, 0.1, 0.0], [0.1, 3.0, 0.2], [0.0, 0.2, 1.0]])), O.satellite(1.3, np.array([[2.0, 0.1, 0.0], [0.1, 3.0, 0.2], [0.0, 0.2, 1.0]])), lambda n: H.rand_rigid(rng, n, 6)),
             (RD.Quadrotor(), O.quadrotor(), lambda n: H.rand_rigid(rng, n, 4, 0.6, 1.9)),
             (RD.Cartpole(), O.cartpole(), lambda n: rng.random((n, 5))),
             (RD.SatellitePair(), O.satellite_pair(), lambda n: rng.random((n, 20)) + 0.1),      # two quaternions: generic projection kernel
             (RD.LieSatellite(), O.lie_satellite(), lambda n: rng.random((n, 10)) + 0.1)]
    for model, om, gen in cases:
        for N in (2, 65, 200):
            Zn = gen(N)
            if Zn.shape[1] in (17, 19):
                Zn[:, 3:7] *= rng.uniform(0.6, 1.6, (N, 1))      # G normalises the quaternion, the dynamics do not
            n, m, e = RD.state_dim(model), RD.control_dim(model), RD.errstate_dim(model)
            t, dt = np.arange(N) * 0.05, np.full(N, 0.05)
            for Q, rule in ((RD.RK4, O.RK4), (RD.RK3, O.RK3)):
                dm = RD.DiscretizedDynamics(model, Q)
                Z = traj(model, Zn, t, dt, "cuda" if where == "cuda" else None)
                E = RD.alloc_error_expansions(dm, N, Z.data)
                y = RD.evaluate(dm, Z)[: N - 1] * 0
                RD.error_expansion_(dm, E, y, Z)
                Eo, yo = om.error_expansion(rule, Zn, t, dt)
                assert host(E).shape == (N - 1, e, e + m)
                np.testing.assert_allclose(host(E), Eo, rtol=0, atol=TOL * max(1.0, np.abs(Eo).max()))
                np.testing.assert_allclose(host(y), yo, rtol=0, atol=TOL)


def test_custom_models_jit_match_the_compiled_in_ones():
    """User-defined dynamics through the hipRTC plug-in: the cartpole and the satellite re-written as source strings must reproduce
    the oracle for every operator (dynamics, RK3/RK4 steps, all three diff methods, rollout, error-state operators, fp32)."""
    rng = np.random.default_rng(31)
    cart = RD.CustomModel(4, 1, H.CARTPOLE_SRC, params=(1.0, 0.2, 0.5, 9.81))
    sat = RD.CustomModel(13, 6, H.SATELLITE_SRC, params=(2.0, 2.0, 3.0, 1.0), liestate=(RD.QuatRotation, (3, 6)))
    for model, om, Zn in ((cart, O.cartpole(), rng.random((150, 5))), (sat, O.body(), H.rand_rigid(rng, 150, 6))):
        N = Zn.shape[0]
        n, m = RD.state_dim(model), RD.control_dim(model)
        t, dt = np.arange(N) * 0.05, np.full(N, 0.05); dt[-1] = 0.0
        Z = traj(model, Zn, t, dt, "cuda")
        np.testing.assert_allclose(host(RD.dynamics(model, Z)), om.dynamics(Zn, t, dt), rtol=0, atol=TOL)
        for Q, rule in ((RD.RK4, O.RK4), (RD.RK3, O.RK3)):
            dm = RD.DiscretizedDynamics(model, Q)
            np.testing.assert_allclose(host(RD.discrete_dynamics(dm, Z)), om.discrete_dynamics(rule, Zn, t, dt), rtol=0, atol=TOL)
            J = RD.alloc_jacobians(dm, N, Z.data)
            Y = torch.empty((N, n), dtype=torch.float64, device="cuda")
            Jo, Yo = om.jacobian(rule, O.FORWARD_AD, Zn, t, dt)
            for diff, tol in ((RD.ForwardAD(), TOL), (RD.UserDefined(), TOL), (RD.FiniteDifference(), 1e-6)):
                J.fill_(3.0)
                RD.jacobian_(RD.StaticReturn(), diff, dm, J, Y, Z)
                np.testing.assert_allclose(host(J), Jo, rtol=0, atol=tol * max(1.0, np.abs(Jo).max()))
                np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        np.testing.assert_allclose(host(RD.dynamics_error(dm, Z)), om.dynamics_error(O.RK4, Zn, t, dt), rtol=0, atol=TOL)
        E = RD.alloc_error_expansions(dm, N, Z.data)
        RD.error_expansion_(dm, E, None, Z)
        np.testing.assert_allclose(host(E), om.error_expansion(O.RK4, Zn, t, dt)[0], rtol=0, atol=TOL * 10)
        G = RD.alloc_errstate_jacobians(model, N, Z.data)
        RD.errstate_jacobian_(model, G, Z, clear=True)
        np.testing.assert_allclose(host(G), om.errstate_jacobian(Zn[:, :n]), rtol=0, atol=TOL)
        # fp32 instantiation of the same source
        Zf = traj(model, Zn, t, dt, "cuda", dtype=np.float32)
        Jf = RD.alloc_jacobians(dm, N, Zf.data)
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, Jf, None, Zf)
        Jo32, _ = om.jacobian(O.RK4, O.FORWARD_AD, Zn.astype(np.float32).astype(np.float64), t, dt)
        np.testing.assert_allclose(host(Jf), Jo32, rtol=1e-4, atol=1e-4)
    # 8-byte aligned slab: custom models with n+m > 10 fall back from the split-role kernel to the element-wise one
    N = 100
    Zn = H.rand_rigid(rng, N, 6); t, dt = np.zeros(N), np.full(N, 0.05)
    Z = traj(sat, Zn, t, dt, "cuda")
    buf = torch.zeros(N * 247 + 1, dtype=torch.float64, device="cuda")
    Ju = buf[1:].view(N, 19, 13).transpose(1, 2)
    for diff, tol in ((RD.ForwardAD(), TOL), (RD.FiniteDifference(), 1e-6), (RD.UserDefined(), TOL)):
        Ju.fill_(5.0)
        RD.jacobian_(RD.StaticReturn(), diff, RD.DiscretizedDynamics(sat, RD.RK4), Ju, None, Z)
        np.testing.assert_allclose(host(Ju), O.body().jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)[0], rtol=0, atol=tol)
    assert float(buf[0]) == 0.0
    # cost-balanced column plan (columns JIT-compiled one by one, code size = cost): same numbers as the contiguous plan
    satb = RD.CustomModel(13, 6, H.SATELLITE_SRC, params=(2.0, 2.0, 3.0, 1.0), liestate=(RD.QuatRotation, (3, 6))).balance_columns()
    Jb = RD.alloc_jacobians(RD.DiscretizedDynamics(satb, RD.RK4), N, Z.data)
    Yb = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), RD.DiscretizedDynamics(satb, RD.RK4), Jb, Yb, Z)
    Jo, Yo = O.body().jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
    np.testing.assert_allclose(host(Jb), Jo, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(Yb), Yo, rtol=0, atol=TOL)
    Eb = RD.alloc_error_expansions(RD.DiscretizedDynamics(satb, RD.RK4), N, Z.data)      # the fused expansion inherits the balanced plan
    RD.error_expansion_(RD.DiscretizedDynamics(satb, RD.RK4), Eb, None, Z)
    np.testing.assert_allclose(host(Eb), O.body().error_expansion(O.RK4, Zn, t, dt)[0], rtol=0, atol=TOL * 10)
    # ImplicitMidpoint on the custom satellite (n = 13: register-matrix Newton kernel, JIT-compiled, LDS kernel gated behind it)
    dmi = RD.DiscretizedDynamics(sat, RD.ImplicitMidpoint)
    Ji = RD.alloc_jacobians(dmi, N, Z.data)
    Yi = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dmi, Ji, Yi, Z)
    Xo, Jio, _ = O.body().implicit_midpoint(Zn, t, dt)
    np.testing.assert_allclose(host(Yi), Xo, rtol=0, atol=1e-9)
    np.testing.assert_allclose(host(Ji), Jio, rtol=0, atol=1e-8 * max(1.0, np.abs(Jio).max()))
    # rollout of the custom cartpole
    M, K = 9, 30
    data = rng.random((M, K, 5)); tt = np.tile(np.arange(K) * 0.02, (M, 1)); dd = np.full((M, K), 0.02)
    stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
    RD.rollout_(RD.StaticReturn(), RD.DiscretizedDynamics(cart, RD.RK4), stack)
    np.testing.assert_allclose(host(stack["data"]), O.cartpole().rollout(O.RK4, data, tt, dd), rtol=0, atol=TOL)


def test_native_gather_c4_shape_on_the_local_ranks():
    """BASELINE configs[3] in the shape the C ABI serves it: Quadrotor errstate_jacobian! on per-rank blocks of knot points, then the stacked G
    through the library's own RCCL exchange (rdhip_comm_create / rdhip_errstate_jacobian_sharded_gather / rdhip_gather) -- no torch.distributed
    in the data path.  The test box has one GPU, so the communicator has one rank per visible device (a 1-rank communicator degenerates to the
    in-stack placement of the local shard and never loads librccl; the driver's 8-GPU node runs the same code with 8 ranks)."""
    import ctypes as C
    from rdhip import _lib as L
    from rdhip import parallel as P
    ndev = min(RD.device_count(), 8)
    comm = P.Comm.local(range(ndev))
    nr, nl = C.c_int(), C.c_int()
    L.check(L.lib().rdhip_comm_size(comm.handle, C.byref(nr), C.byref(nl)))
    assert (nr.value, nl.value) == (ndev, ndev)
    rng = np.random.default_rng(4)
    model, om = RD.Quadrotor(), O.quadrotor()
    N = 4099                                                        # ragged: the last block is short
    Zn = H.rand_rigid(rng, N, 4, 0.6, 1.9)
    Xn = np.ascontiguousarray(Zn[:, :13])
    Go = om.errstate_jacobian(Xn)
    blocks = [P.shard_range(N, r, ndev) for r in range(ndev)]
    for mode, code in (("root", L.GATHER_ROOT), ("all", L.GATHER_ALL), ("none", L.GATHER_NONE)):
        Xs, Gs, stacks = [], [], []
        for r, (lo, hi) in enumerate(blocks):
            dev = f"cuda:{r}"
            Xs.append(torch.as_tensor(Xn[lo:hi]).to(dev))
            Gs.append(torch.full((hi - lo, 12, 13), 7.0, dtype=torch.float64, device=dev))
            stacks.append(torch.full((N, 12, 13), 7.0, dtype=torch.float64, device=dev))
        torch.cuda.synchronize()
        arr = lambda ts: (C.c_void_p * ndev)(*[t.data_ptr() for t in ts])      # noqa: E731
        rc = L.lib().rdhip_errstate_jacobian_sharded_gather(comm.handle, model._handle, L.F64, N, arr(Xs), 13, 1, arr(Gs), code, 0, arr(stacks))
        L.check(rc, comm.ctxs[0].handle)
        comm.synchronize()
        for r, (lo, hi) in enumerate(blocks):                       # every rank holds its own block
            np.testing.assert_allclose(host(Gs[r]).transpose(0, 2, 1), Go[lo:hi], rtol=0, atol=TOL)
        for r in range(ndev):
            got = host(stacks[r]).transpose(0, 2, 1)
            if mode == "none" or (mode == "root" and r != 0):
                assert np.all(got == 7.0)                           # untouched
            else:
                np.testing.assert_allclose(got, Go, rtol=0, atol=TOL)
    # Jacobian shards computed IN PLACE inside the root's stack: no copy for the root's own block
    sat, osat = RD.Satellite(), O.satellite()
    dm = RD.DiscretizedDynamics(sat, RD.RK4)
    Zn = H.rand_rigid(rng, N, 6); t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
    Jo, _ = osat.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
    Zs, ts, dts, Js = [], [], [], []
    stack0 = torch.zeros((N, 19, 13), dtype=torch.float64, device="cuda:0")
    for r, (lo, hi) in enumerate(blocks):
        dev = f"cuda:{r}"
        Zs.append(torch.as_tensor(Zn[lo:hi]).to(dev)); ts.append(torch.as_tensor(t[lo:hi]).to(dev)); dts.append(torch.as_tensor(dt[lo:hi]).to(dev))
        Js.append(stack0[lo:hi] if r == 0 else torch.zeros((hi - lo, 19, 13), dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    stacks = [stack0] + [torch.zeros(1, dtype=torch.float64, device=f"cuda:{r}") for r in range(1, ndev)]
    rc = L.lib().rdhip_jacobian_sharded_gather(comm.handle, sat._handle, L.RK4, L.FORWARD_AD, L.F64, N, arr(Zs), arr(ts), arr(dts), arr(Js), None, L.GATHER_ROOT, 0, arr(stacks))
    L.check(rc, comm.ctxs[0].handle)
    comm.synchronize()
    np.testing.assert_allclose(host(stack0).transpose(0, 2, 1), Jo, rtol=0, atol=TOL)
    # the per-process form with one rank: rdhip_comm_create_rank(ctx, 1, 0, id)
    c1 = P.Comm.from_torch_distributed(0)
    out = torch.zeros(16, dtype=torch.float64, device="cuda:0")
    src = torch.arange(16, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    c1.gather("all", [src], [16 * 8], [out])
    c1.synchronize()
    assert torch.equal(out, src)
    with pytest.raises(ValueError):
        c1.gather("root", [src], [16 * 8], [out], root=3)


def test_baseline_config_sizes_c1_c2():
    """BASELINE.json configs[0] and configs[1] at their exact sizes.  C1: Cartpole (n=4, m=1), RK3 discrete_dynamics + discrete Jacobian on ONE
    trajectory of 101 knot points (examples/cartpole.jl:46: dt = 0.1; restated with the v0.4 calls), through the v0.3 spellings
    too.  C2: Cartpole batched RK4 discrete Jacobian, 4 096 knot points, fp64."""
    rng = np.random.default_rng(1)
    model, om = RD.Cartpole(), O.cartpole()
    # C1
    N = 101
    Zn = rng.random((N, 5)); t = np.arange(N) * 0.1; dt = np.full(N, 0.1); dt[-1] = 0.0
    for where in ("host", "cuda"):
        Z = traj(model, Zn, t, dt, "cuda" if where == "cuda" else None)
        dm = RD.DiscretizedDynamics[RD.RK3](model)
        np.testing.assert_allclose(host(RD.discrete_dynamics(dm, Z)), om.discrete_dynamics(O.RK3, Zn, t, dt), rtol=0, atol=TOL)
        J = RD.alloc_jacobians(dm, N, Z.data)
        y = RD.evaluate(dm, Z) * 0
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, y, Z)
        Jo, Yo = om.jacobian(O.RK3, O.FORWARD_AD, Zn, t, dt)
        np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(y), Yo, rtol=0, atol=TOL)
        J3 = RD.alloc_jacobians(dm, N, Z.data)
        RD.discrete_jacobian_(RD.RK3, J3, model, Z)                                  # v0.3: discrete_jacobian!(RK3, D, model, Z)
        np.testing.assert_allclose(host(J3), Jo, rtol=0, atol=TOL)
        z0 = RD.KnotPoint(Zn[0, :4], Zn[0, 4:], 0.0, 0.1)                            # the per-knot call of the reference on the same knot point
        np.testing.assert_allclose(RD.discrete_dynamics(dm, z0), om.discrete_dynamics(O.RK3, Zn[:1], t[:1], dt[:1])[0], rtol=0, atol=TOL)
    # C2
    N = 4096
    rng = np.random.default_rng(2)
    Zn = rng.random((N, 5)); t = np.arange(N) * 0.1; dt = np.full(N, 0.1)
    Z = traj(model, Zn, t, dt, "cuda")
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    J = RD.alloc_jacobians(dm, N, Z.data)
    Y = torch.empty((N, 4), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
    Jo, Yo = om.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
    np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)


def test_jacobian_multi_fuses_independent_trajectories_into_one_launch():
    """rdhip_jacobian_multi: several independent packed trajectories (different arrays, ragged lengths) through ONE persistent tile loop must
    give exactly what one rdhip_jacobian call per trajectory gives -- and both must match the oracle.  Covers the split-role kernel (satellite,
    quadrotor), the all-columns-per-lane kernel (cartpole), > 32 trajectories (the device table: ragged lengths -> binary search, equal
    lengths -> no search), empty trajectories, optional f(z) outputs, FiniteDifference / UserDefined / unaligned slabs (fused since round 3), and
    the combinations without a fused kernel (implicit rule, host memory), which run batch by batch."""
    rng = np.random.default_rng(123)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    cases = ((RD.Satellite(), O.satellite(), lambda N: H.rand_rigid(rng, N, 6), [4096, 101, 64, 1, 65, 0, 777]),
             (RD.Satellite(), O.satellite(), lambda N: H.rand_rigid(rng, N, 6), [101] * 70),                        # equal lengths, device table
             (RD.Satellite(), O.satellite(), lambda N: H.rand_rigid(rng, N, 6), [37, 0, 129] * 13 + [64]),          # ragged, device table, empty batches
             (RD.Quadrotor(), O.quadrotor(), lambda N: H.rand_rigid(rng, N, 4, 0.6, 1.9), [130, 63]),
             (RD.Cartpole(), O.cartpole(), lambda N: rng.random((N, 5)), [101] * 40 + [4096, 1, 0, 65]))
    for ci, (model, om, gen, sizes) in enumerate(cases):
        n, m = RD.state_dim(model), RD.control_dim(model)
        for Q, rule in ((RD.RK4, O.RK4), (RD.RK3, O.RK3)):
            if ci in (1, 2) and Q is RD.RK3:
                continue
            dm = RD.DiscretizedDynamics(model, Q)
            Zn = [gen(N) for N in sizes]
            ts = [0.1 * b + np.arange(N) * 0.05 for b, N in enumerate(sizes)]
            dts = [np.full(N, 0.05) for N in sizes]
            for d in dts:
                if len(d):
                    d[-1] = 0.0                                                   # every trajectory ends in a terminal knot point
            Zs = [traj(model, z, t, d, "cuda") for z, t, d in zip(Zn, ts, dts)]
            Js = [RD.alloc_jacobians(dm, N, Zs[0].data).fill_(7.0) for N in sizes]
            Ys = [torch.full((N, n), 7.0, dtype=torch.float64, device="cuda") if b % 3 else None for b, N in enumerate(sizes)]
            assert RD.jacobian_multi_(sig, AD, dm, Js, Ys, Zs) == 1              # ONE launch
            for b, N in enumerate(sizes):
                if N == 0:
                    continue
                Jo, Yo = om.jacobian(rule, O.FORWARD_AD, Zn[b], ts[b], dts[b])
                np.testing.assert_allclose(host(Js[b]), Jo, rtol=0, atol=TOL)
                if Ys[b] is not None:
                    np.testing.assert_allclose(host(Ys[b]), Yo, rtol=0, atol=TOL)
                if ci in (1, 2) and b > 4:
                    continue
                J1 = RD.alloc_jacobians(dm, N, Zs[b].data)
                RD.jacobian_(sig, AD, dm, J1, None, Zs[b])
                # the same arithmetic per column as the single-batch launch; batches of <= 8 192 knot points take the column-split kernel there
                # (a separately compiled instantiation of the same column code: FMA contraction may differ in the last bit), larger ones the
                # very same tile loop
                # (small models: the single- and the multi-batch kernel are two instantiations of the same code as well -- they agreed bit for bit
                # until an unrelated change to the translation unit moved the compiler's inlining: held to the last bits, not to equality)
                if N > 8192:
                    assert torch.equal(J1, Js[b])
                else:
                    np.testing.assert_allclose(host(J1), host(Js[b]), rtol=0, atol=1e-13)
    # FiniteDifference, the UserDefined chain rule and unaligned slabs: NO batch-table kernel since round 6 (their instantiations were half of the library's
    # code and nothing tested them: csrc/dispatch.hpp launch_jacobian_multi) -- four batches and more are gathered into one stack, run as ONE rdhip_jacobian
    # launch and scattered (three launches; fewer batches: one launch each); bit-identical to the per-trajectory calls; > 32 trajectories of a rigid body
    # and of the cartpole
    for model, om, gen in ((RD.Satellite(), O.satellite(), lambda N: H.rand_rigid(rng, N, 6)), (RD.Cartpole(), O.cartpole(), lambda N: rng.random((N, 5)))):
        n, m = RD.state_dim(model), RD.control_dim(model)
        sizes = [70, 33, 0, 101] + [65] * 31
        Zn = [gen(N) for N in sizes]; ts = [np.arange(N) * 0.05 for N in sizes]; dts = [np.full(N, 0.05) for N in sizes]
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        Zs = [traj(model, z, t, d, "cuda") for z, t, d in zip(Zn, ts, dts)]
        for diff, odiff, tol in ((RD.FiniteDifference(), O.FORWARD_AD, 1e-6), (RD.UserDefined(), O.USER, TOL)):
            Js = [RD.alloc_jacobians(dm, N, Zs[0].data).fill_(7.0) for N in sizes]
            Ys = [torch.full((N, n), 7.0, dtype=torch.float64, device="cuda") for N in sizes]
            assert RD.jacobian_multi_(sig, diff, dm, Js, Ys, Zs) == 3        # gather + ONE rdhip_jacobian over the 34 batches' concatenation + scatter (api.hip multi_staged)
            for b in (0, 1, 3, 4, 34):
                Jo, Yo = om.jacobian(O.RK4, odiff, Zn[b], ts[b], dts[b])
                np.testing.assert_allclose(host(Js[b]), Jo, rtol=0, atol=tol)
                np.testing.assert_allclose(host(Ys[b]), Yo, rtol=0, atol=TOL)
                J1 = RD.alloc_jacobians(dm, sizes[b], Zs[b].data)
                RD.jacobian_(sig, diff, dm, J1, None, Zs[b])
                assert torch.equal(J1, Js[b])
    model, om = RD.Satellite(), O.satellite()
    sizes = [70, 33]
    Zn = [H.rand_rigid(rng, N, 6) for N in sizes]; ts = [np.arange(N) * 0.05 for N in sizes]; dts = [np.full(N, 0.05) for N in sizes]
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    Zs = [traj(model, z, t, d, "cuda") for z, t, d in zip(Zn, ts, dts)]
    Js = [RD.alloc_jacobians(dm, N, Zs[0].data) for N in sizes]
    buf = torch.zeros(sizes[0] * 247 + 1, dtype=torch.float64, device="cuda")
    Ju = [buf[1:].view(sizes[0], 19, 13).transpose(1, 2), Js[1]]                  # 8-byte aligned slab: batch by batch (the element-wise copy kernels)
    assert RD.jacobian_multi_(sig, AD, dm, Ju, None, Zs) == 2
    for b in range(2):
        np.testing.assert_allclose(host(Ju[b]), om.jacobian(O.RK4, O.FORWARD_AD, Zn[b], ts[b], dts[b])[0], rtol=0, atol=TOL)
    assert float(buf[0]) == 0.0
    # no fused kernel: ImplicitMidpoint, host arrays -> batch by batch, same answers, and the call says so
    dmi = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
    Ys = [torch.empty((N, 13), dtype=torch.float64, device="cuda") for N in sizes]
    assert RD.jacobian_multi_(sig, RD.ImplicitFunctionTheorem(AD), dmi, Js, Ys, Zs) == 3 * len(sizes)
    for b in range(2):
        Xo, Jio, _ = om.implicit_midpoint(Zn[b], ts[b], dts[b])
        np.testing.assert_allclose(host(Ys[b]), Xo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(host(Js[b]), Jio, rtol=0, atol=1e-8 * max(1.0, np.abs(Jio).max()))
    Zh = [traj(model, z, t, d) for z, t, d in zip(Zn, ts, dts)]
    Jh = [RD.alloc_jacobians(dm, N, Zh[0].data) for N in sizes]
    assert RD.jacobian_multi_(sig, AD, dm, Jh, None, Zh) == len(sizes)
    for b in range(2):
        np.testing.assert_allclose(host(Jh[b]), om.jacobian(O.RK4, O.FORWARD_AD, Zn[b], ts[b], dts[b])[0], rtol=0, atol=TOL)
    # a capturing context with more trajectories than fit the kernel arguments: groups of 32 launches (nothing may be allocated inside a capture)
    from rdhip import _lib as L
    sizes = [64] * 40
    Zn = [H.rand_rigid(rng, N, 6) for N in sizes]; ts = [np.arange(N) * 0.05 for N in sizes]; dts = [np.full(N, 0.05) for N in sizes]
    s1 = torch.cuda.Stream()
    with torch.cuda.stream(s1):
        Zs = [traj(model, z, t, d, "cuda") for z, t, d in zip(Zn, ts, dts)]
        Js = [RD.alloc_jacobians(dm, N, Zs[0].data).zero_() for N in sizes]
        call = RD.prepare_jacobian_multi(sig, AD, dm, Js, None, Zs)
        call()
        assert call.launches == 1
        torch.cuda.synchronize()
        ref = [j.clone() for j in Js]
        for j in Js:
            j.zero_()
        ctx = L.context_for(0, s1.cuda_stream)
        ctx.begin_capture()
        call()
        g = ctx.end_capture()
        assert call.launches == 2
        ctx.launch(g)
        torch.cuda.synchronize()
        for a, b in zip(ref, Js):
            assert torch.equal(a, b)


def test_time_dependent_dynamics_receive_the_stage_times():
    """dynamics(model, x, u, t): the time argument reaches every evaluation the way the reference passes it -- the knot point's t for the
    continuous model, t / t + h/2 / t + h inside Euler, RK2, RK3, RK4 (src/integration.jl:73-83,130-135,280-286) and in the chain rule's
    stage Jacobians (:149-177,:302-337), t + h/2 in the implicit-midpoint residual and t in its Jacobian (:640,:684).  The reference's own
    models are time-invariant, so the check runs two explicitly time-dependent user-defined (hipRTC) models against the oracle's twins;
    n + m = 3 takes the all-columns-per-lane kernels, n + m = 19 the split-role compile-time-column ones."""
    rng = np.random.default_rng(77)
    pend = RD.CustomModel(2, 1, H.TV_PENDULUM_SRC, params=(9.81, 2.0, 3.0, 0.1))
    sat = RD.CustomModel(13, 6, H.TV_SATELLITE_SRC, params=(2.0, 2.0, 3.0, 1.0, 0.7, 0.4, 2.5), liestate=(RD.QuatRotation, (3, 6)))
    for model, om, Zn in ((pend, O.tv_pendulum(), rng.random((200, 3))), (sat, O.tv_satellite(), H.rand_rigid(rng, 150, 6))):
        N = Zn.shape[0]
        n, m = RD.state_dim(model), RD.control_dim(model)
        t, dt = 0.3 + np.arange(N) * 0.13, np.full(N, 0.07); dt[-1] = 0.0
        Z = traj(model, Zn, t, dt, "cuda")
        ref = om.dynamics(Zn, t, dt)
        assert np.abs(ref - om.dynamics(Zn, 0 * t, dt)).max() > 1e-2             # the case does depend on t
        np.testing.assert_allclose(host(RD.dynamics(model, Z)), ref, rtol=0, atol=TOL)
        for Q, rule in ((RD.Euler, O.EULER), (RD.RK2, O.RK2), (RD.RK3, O.RK3), (RD.RK4, O.RK4)):
            dm = RD.DiscretizedDynamics(model, Q)
            np.testing.assert_allclose(host(RD.discrete_dynamics(dm, Z)), om.discrete_dynamics(rule, Zn, t, dt), rtol=0, atol=TOL)
            J = RD.alloc_jacobians(dm, N, Z.data)
            Y = torch.empty((N, n), dtype=torch.float64, device="cuda")
            Jo, Yo = om.jacobian(rule, O.FORWARD_AD, Zn, t, dt)
            for diff, tol in ((RD.ForwardAD(), TOL), (RD.UserDefined(), TOL), (RD.FiniteDifference(), 1e-6)):
                J.fill_(3.0)
                RD.jacobian_(RD.StaticReturn(), diff, dm, J, Y, Z)
                np.testing.assert_allclose(host(J), Jo, rtol=0, atol=tol * max(1.0, np.abs(Jo).max()))
                np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        np.testing.assert_allclose(host(RD.dynamics_error(dm, Z)), om.dynamics_error(O.RK4, Zn, t, dt), rtol=0, atol=TOL)
        if model is sat:                                                            # fused error-state expansion with a time-dependent model
            E = RD.alloc_error_expansions(dm, N, Z.data)
            RD.error_expansion_(dm, E, None, Z)
            np.testing.assert_allclose(host(E), om.error_expansion(O.RK4, Zn, t, dt)[0], rtol=0, atol=TOL * 10)
        # ImplicitMidpoint: Newton solve, IFT Jacobian, dynamics error and its Jacobians
        dmi = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
        Ji = RD.alloc_jacobians(dmi, N, Z.data)
        Yi = torch.empty((N, n), dtype=torch.float64, device="cuda")
        RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dmi, Ji, Yi, Z)
        Xo, Jio, _ = om.implicit_midpoint(Zn, t, dt)
        np.testing.assert_allclose(host(Yi), Xo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(host(Ji), Jio, rtol=0, atol=1e-8 * max(1.0, np.abs(Jio).max()))
        np.testing.assert_allclose(host(RD.discrete_dynamics(dmi, Z)), Xo, rtol=0, atol=1e-9)
        Eo, J2o, J1o = om.implicit_midpoint_error(Zn, t, dt)
        np.testing.assert_allclose(host(RD.dynamics_error(dmi, Z)), Eo, rtol=0, atol=TOL)
        J2 = RD.alloc_jacobians(dmi, N - 1, Z.data); J1 = RD.alloc_jacobians(dmi, N - 1, Z.data)
        RD.dynamics_error_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dmi, J2, J1, None, None, Z)
        np.testing.assert_allclose(host(J2), J2o, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(J1), J1o, rtol=0, atol=TOL)
        # rollout: knot k advances with ITS time stamp
        M, K = 7, 25
        data = (rng.random((M, K, n + m)) if n == 2 else H.rand_rigid(rng, M * K, m).reshape(M, K, n + m))
        tt = 0.2 + np.tile(np.arange(K) * 0.05, (M, 1)) + np.arange(M)[:, None] * 0.31
        dd = np.full((M, K), 0.05)
        for Q, rule in ((RD.RK4, O.RK4), (RD.RK3, O.RK3)):
            stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
            RD.rollout_(RD.StaticReturn(), RD.DiscretizedDynamics(model, Q), stack)
            np.testing.assert_allclose(host(stack["data"]), om.rollout(rule, data, tt, dd), rtol=0, atol=TOL)
        stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
        RD.rollout_(RD.StaticReturn(), dmi, stack)                                  # implicit rollout: a Newton solve per step at that step's time
        Zr = data.copy()
        for k in range(K - 1):
            Zr[:, k + 1, :n] = om.implicit_midpoint(Zr[:, k, :], tt[:, k], dd[:, k], want_jacobian=False)[0]
        np.testing.assert_allclose(host(stack["data"]), Zr, rtol=0, atol=1e-8)


@pytest.mark.parametrize("name", ["cartpole", "satellite_world", "body_body", "satfull_world", "quadrotor_world", "satpair", "liesat"])
@pytest.mark.parametrize("where", ["host", "cuda"])
def test_golden_error_expansion_and_dynamics_error(name, where):
    """HIP path against the definition-derived fixtures exp_*.npz (torch, independent of both the oracle and the L(q)H closed form)."""
    d, g = H.load("dyn_" + name), H.load("exp_" + name)
    model = rd_model(name, d)
    N = d["Z"].shape[0]
    Z = traj(model, d["Z"], d["t"], d["dt"], "cuda" if where == "cuda" else None)
    for rn, Q in (("rk3", RD.RK3), ("rk4", RD.RK4)):
        dm = RD.DiscretizedDynamics(model, Q)
        np.testing.assert_allclose(host(RD.dynamics_error(dm, Z)), g["err_" + rn], rtol=0, atol=TOL)
        E = RD.alloc_error_expansions(dm, N, Z.data)
        RD.error_expansion_(dm, E, None, Z)
        np.testing.assert_allclose(host(E), g["E_" + rn], rtol=0, atol=TOL * max(1.0, np.abs(g["E_" + rn]).max()))


@pytest.mark.parametrize("where", ["host", "cuda"])
def test_implicit_midpoint_newton_solve_and_ift_jacobian(where):
    """DiscretizedDynamics{ImplicitMidpoint} (src/integration.jl:620-729; Newton :422-463; IFT Jacobian :517-538; test/implicit_dynamics_test.jl):
    next state, IFT Jacobian, dynamics error and its Jacobians against the oracle; the solution satisfies the implicit equation, and
    the dynamics error of (z1, [x_next; u]) vanishes (test/implicit_dynamics_test.jl:46-60)."""
    rng = np.random.default_rng(41)
    cases = [(RD.Cartpole(), O.cartpole(), lambda n: rng.random((n, 5))), (RD.Satellite(), O.satellite(), lambda n: H.rand_rigid(rng, n, 6)),
             (RD.Quadrotor(), O.quadrotor(), lambda n: H.rand_rigid(rng, n, 4, 0.6, 1.9)), (RD.DoubleIntegrator(2), O.double_integrator(2), lambda n: rng.random((n, 6))),
             (RD.SatellitePair(), O.satellite_pair(), lambda n: rng.random((n, 20)) + 0.1)]
    for model, om, gen in cases:
        for N in (1, 70):
            Zn = gen(N)
            n, m = RD.state_dim(model), RD.control_dim(model)
            t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
            if N > 2:
                dt[-1] = 0.0
            dm = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
            assert isinstance(RD.default_diffmethod(dm), RD.ImplicitFunctionTheorem)
            Z = traj(model, Zn, t, dt, "cuda" if where == "cuda" else None)
            Xo, Jo, its = om.implicit_midpoint(Zn, t, dt)
            assert its.max() <= 6
            Xn = RD.discrete_dynamics(dm, Z)
            np.testing.assert_allclose(host(Xn), Xo, rtol=0, atol=TOL)
            J = RD.alloc_jacobians(dm, N, Z.data)
            Y = RD.evaluate(dm, Z) * 0
            RD.jacobian_(RD.StaticReturn(), RD.default_diffmethod(dm), dm, J, Y, Z)
            np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL * max(1.0, np.abs(Jo).max()))
            np.testing.assert_allclose(host(Y), Xo, rtol=0, atol=TOL)
            with pytest.raises(ValueError):                    # ArgumentError in the reference (src/discretized_dynamics.jl:243-249)
                RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
            # the solution satisfies x1 + h f((x1 + x2)/2, u) - x2 = 0
            xm = np.concatenate([(Zn[:, :n] + host(Xn)) / 2, np.where(dt[:, None] == 0, 0.0, Zn[:, n:])], axis=1)
            res = Zn[:, :n] + dt[:, None] * om.dynamics(xm, t, np.full(N, 0.1)) - host(Xn)
            assert np.abs(res).max() < 1e-11
            if N > 1:
                Eo, J2o, J1o = om.implicit_midpoint_error(Zn, t, dt)
                np.testing.assert_allclose(host(RD.dynamics_error(dm, Z)), Eo, rtol=0, atol=TOL)
                J2, J1 = RD.alloc_jacobians(dm, N - 1, Z.data), RD.alloc_jacobians(dm, N - 1, Z.data)
                y2 = RD.evaluate(dm, Z)[: N - 1] * 0
                RD.dynamics_error_jacobian_(RD.StaticReturn(), RD.default_diffmethod(dm), dm, J2, J1, y2, None, Z)
                np.testing.assert_allclose(host(J2), J2o, rtol=0, atol=TOL * max(1.0, np.abs(J2o).max()))
                np.testing.assert_allclose(host(J1), J1o, rtol=0, atol=TOL * max(1.0, np.abs(J1o).max()))
                np.testing.assert_allclose(host(y2), Eo, rtol=0, atol=TOL)
    # rollout!(sig, DiscretizedDynamics{ImplicitMidpoint}(model), Z, x0): repeated Newton solves along each trajectory
    for model, om, nz in ((RD.Cartpole(), O.cartpole(), 5), (RD.Satellite(), O.satellite(), 19)):
        n = RD.state_dim(model)
        M, K = 70, 12
        data = (H.rand_rigid(rng, M * K, 6) if nz == 19 else rng.random((M * K, nz))).reshape(M, K, nz)
        tt, dd = np.tile(np.arange(K) * 0.05, (M, 1)), np.full((M, K), 0.05)
        ref = data.copy()
        for k in range(1, K):
            ref[:, k, :n] = om.implicit_midpoint(ref[:, k - 1], tt[:, k - 1], dd[:, k - 1], want_jacobian=False)[0]
        stack = {"data": data.copy(), "t": tt, "dt": dd}
        if where == "cuda":
            stack = {k_: torch.as_tensor(v).cuda() for k_, v in stack.items()}
        RD.rollout_(RD.StaticReturn(), RD.DiscretizedDynamics(model, RD.ImplicitMidpoint), stack)
        np.testing.assert_allclose(host(stack["data"]), ref, rtol=0, atol=TOL)
    # a JIT-compiled custom model under the implicit rule
    cm = RD.CustomModel(4, 1, H.CARTPOLE_SRC, params=(1.0, 0.2, 0.5, 9.81))
    Zc = rng.random((40, 5)); tc, dc = np.zeros(40), np.full(40, 0.1)
    dmc = RD.DiscretizedDynamics(cm, RD.ImplicitMidpoint)
    Zt = traj(cm, Zc, tc, dc, "cuda" if where == "cuda" else None)
    Xo, Jo, _ = O.cartpole().implicit_midpoint(Zc, tc, dc)
    Jc = RD.alloc_jacobians(dmc, 40, Zt.data)
    RD.jacobian_(RD.StaticReturn(), RD.default_diffmethod(dmc), dmc, Jc, None, Zt)
    np.testing.assert_allclose(host(Jc), Jo, rtol=0, atol=TOL * max(1.0, np.abs(Jo).max()))
    np.testing.assert_allclose(host(RD.discrete_dynamics(dmc, Zt)), Xo, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(RD.dynamics_error(dmc, Zt)), O.cartpole().implicit_midpoint_error(Zc, tc, dc)[0], rtol=0, atol=TOL)
    # fewer Newton iterations than needed: the iterate after 1 step, as the oracle with newton_iters = 1
    model, om = RD.Cartpole(), O.cartpole()
    RD.setnewtoniters(model, 1)
    Zn = rng.random((9, 5)); t, dt = np.zeros(9), np.full(9, 0.1)
    dm = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
    np.testing.assert_allclose(host(RD.discrete_dynamics(dm, traj(model, Zn, t, dt, "cuda" if where == "cuda" else None))), om.implicit_midpoint(Zn, t, dt, newton_iters=1)[0], rtol=0, atol=TOL)


def test_implicit_midpoint_lu_with_row_exchanges():
    """The Newton matrix (h/2) A - I of the physical models is diagonally dominant, so their LU never exchanges rows; a linear system with a
    large step makes partial pivoting do real work (checked: scipy's LU of the same matrix permutes), and its solution and IFT Jacobian have
    closed forms: x2 = (I - hA/2)^-1 ((I + hA/2) x1 + h B u), d x2 / d x1 = (I - hA/2)^-1 (I + hA/2), d x2 / d u = h (I - hA/2)^-1 B."""
    import scipy.linalg
    rng = np.random.default_rng(35)
    n, m, h = 6, 3, 1.0
    A = 4.0 * rng.standard_normal((n, n)); np.fill_diagonal(A, 2.0 / h + 1e-3 * rng.standard_normal(n))     # (h/2) A - I has a tiny diagonal
    B = rng.standard_normal((n, m))
    Mn = 0.5 * h * A - np.eye(n)
    perm = scipy.linalg.lu(Mn)[0]
    assert not np.allclose(perm, np.eye(n))                  # the factorisation really exchanges rows
    model, om = RD.LinearModel(A, B), O.linear(A, B)
    dm = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
    N = 200
    Zn = rng.standard_normal((N, n + m)); t, dt = np.zeros(N), np.full(N, h)
    Minv = np.linalg.inv(np.eye(n) - 0.5 * h * A)
    Xc = (Minv @ ((np.eye(n) + 0.5 * h * A) @ Zn[:, :n].T + h * B @ Zn[:, n:].T)).T
    Jc = np.concatenate([Minv @ (np.eye(n) + 0.5 * h * A), h * Minv @ B], axis=1)
    for where in ("cuda", None):
        Z = traj(model, Zn, t, dt, where)
        J = RD.alloc_jacobians(dm, N, Z.data)
        Y = torch.empty((N, n), dtype=torch.float64, device="cuda") if where else np.empty((N, n))
        RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dm, J, Y, Z)
        scale = max(1.0, np.abs(Jc).max())
        np.testing.assert_allclose(host(Y), Xc, rtol=0, atol=1e-9 * max(1.0, np.abs(Xc).max()))
        np.testing.assert_allclose(host(J), np.broadcast_to(Jc, (N, n, n + m)), rtol=0, atol=1e-9 * scale)
        np.testing.assert_allclose(host(RD.discrete_dynamics(dm, Z)), Xc, rtol=0, atol=1e-9 * max(1.0, np.abs(Xc).max()))
    Xo, Jo, _ = om.implicit_midpoint(Zn, t, dt)
    np.testing.assert_allclose(Xo, Xc, rtol=0, atol=1e-9 * max(1.0, np.abs(Xc).max()))


def test_implicit_midpoint_register_pass_falls_back_when_rows_must_be_exchanged():
    """Models with n > 6 run the Newton solve with the matrix in registers and WITHOUT row exchanges, check per lane that partial pivoting
    would not have exchanged rows, and otherwise raise a flag that makes the LDS kernels (real partial pivoting) recompute the batch.  A
    satellite with a very large step (h = 40: (h/2) A - I far from diagonally dominant) forces that path for some knot points; the
    results must still be the oracle's (which pivots), for the solve, the IFT Jacobian and the rollout, next to well-behaved knots."""
    rng = np.random.default_rng(36)
    model, om = RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0])), O.body()
    dm = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
    RD.setnewtoniters(dm, 4)                                  # the wild knots need not converge: same iterates either way
    N = 300
    Zn = H.rand_rigid(rng, N, 6); Zn[:, 10:13] *= 3.0
    t = np.zeros(N); dt = np.full(N, 0.05); dt[::7] = 40.0
    Xo, Jo, _ = om.implicit_midpoint(Zn, t, dt, newton_iters=4)
    assert np.all(np.isfinite(Xo)) and np.all(np.isfinite(Jo))
    for where in ("cuda", None):
        Z = traj(model, Zn, t, dt, where)
        J = RD.alloc_jacobians(dm, N, Z.data)
        Y = torch.empty((N, 13), dtype=torch.float64, device="cuda") if where else np.empty((N, 13))
        RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dm, J, Y, Z)
        sx, sj = max(1.0, np.abs(Xo).max()), max(1.0, np.abs(Jo).max())
        np.testing.assert_allclose(host(Y), Xo, rtol=0, atol=1e-9 * sx)
        np.testing.assert_allclose(host(J), Jo, rtol=0, atol=1e-8 * sj)
        np.testing.assert_allclose(host(RD.discrete_dynamics(dm, Z)), Xo, rtol=0, atol=1e-9 * sx)
    RD.setnewtoniters(dm, 10)


def test_implicit_midpoint_jacobian_at_the_last_tested_iterate_when_the_iteration_limit_stops_the_solve():
    """The reference takes the IFT Jacobian where its Newton loop last evaluated the residual (src/integration.jl:422-463, :517-538); when the
    iteration limit ends the loop, x_next is one update past that point.  The tile kernel (implicit_ift_split_kernel) linearises about x_next,
    so knots that stopped at the limit must be routed to the exact kernel: with 2 iterations and tol 1e-12 no ordinary knot converges, a
    terminal knot (h = 0, residual 0 at the first test) does, and both kinds sit in the same tiles."""
    rng = np.random.default_rng(52)
    for model, om, mctl in ((RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0])), O.body(), 6), (RD.Quadrotor(), O.quadrotor(), 4)):
        dm = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
        N = 64 * 5 + 9
        Zn = H.rand_rigid(rng, N, mctl, 0.6 if mctl == 4 else 0.0, 1.9 if mctl == 4 else 1.0)
        t = np.arange(N) * 0.1; dt = np.full(N, 0.1); dt[::5] = 0.0
        try:
            for iters in (2, 10):
                RD.setnewtoniters(dm, iters)
                Xo, Jo, its = om.implicit_midpoint(Zn, t, dt, newton_iters=iters)
                Z = traj(model, Zn, t, dt, "cuda")
                J = RD.alloc_jacobians(dm, N, Z.data)
                Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
                RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dm, J, Y, Z)
                np.testing.assert_allclose(host(Y), Xo, rtol=0, atol=TOL)
                np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL)
        finally:
            RD.setnewtoniters(dm, 10)


def test_rollout_matches_oracle_and_is_idempotent():
    rng = np.random.default_rng(9)
    M, K = 37, 21
    model, om = RD.Cartpole(), O.cartpole()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    data = rng.random((M, K, 5))
    t = np.broadcast_to(np.arange(K) * 0.05, (M, K)).copy()
    dt = np.full((M, K), 0.05); dt[:, -1] = 0.0
    ref = om.rollout(O.RK4, data, t, dt)
    stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(t).cuda(), "dt": torch.as_tensor(dt).cuda()}
    RD.rollout_(RD.StaticReturn(), dm, stack)
    np.testing.assert_allclose(host(stack["data"]), ref, rtol=0, atol=TOL)
    RD.rollout_(RD.InPlace(), dm, stack)      # rolling out an already rolled-out trajectory changes nothing
    np.testing.assert_allclose(host(stack["data"]), ref, rtol=0, atol=TOL)
    # single trajectory through the SampledTrajectory-style constructor, with x0
    X, U = rng.random((K, 4)), rng.random((K - 1, 1))
    Zt = RD.PackedTrajectory.from_states_controls(X, U, dt=0.05)
    x0 = rng.random(4)
    RD.rollout_(RD.StaticReturn(), dm, Zt, x0)
    ref1 = om.rollout(O.RK4, np.concatenate([X, np.vstack([U, np.zeros((1, 1))])], axis=1)[None], Zt.t[None], Zt.dt[None], X0=x0[None])
    np.testing.assert_allclose(Zt.data, ref1[0], rtol=0, atol=TOL)


@pytest.mark.parametrize("variant", ["satellite_world", "satellite_body", "satellite_fullJ_world", "quadrotor_world", "quadrotor_body"])
def test_cascade_finite_difference_matches_the_full_evaluation(variant):
    """FiniteDifference for the rigid bodies integrates, per column, only the cascade blocks the perturbed component can change (csrc/cascade.hpp).
    It must reproduce forward differences with full evaluations -- the oracle's, src/jacobian_gen.jl:643-685 -- for every explicit rule, both
    velocity frames, diagonal and full inertia, a terminal knot point (zero control columns) and a ragged last tile; blocks a column cannot
    reach are exact zeros in both; and it stays within the reference's 1e-6 of the exact Jacobian (test/integration_tests.jl:14)."""
    rng = np.random.default_rng(77)
    body = variant.endswith("_body")
    if variant.startswith("satellite"):
        Jm = np.array([[2.0, 0.1, 0.0], [0.1, 3.0, 0.2], [0.0, 0.2, 1.0]]) if "fullJ" in variant else np.diag([2.0, 3.0, 1.0])
        model, om, mctl = RD.Satellite(mass=2.0, J=Jm, bodyframe=body), O.satellite(2.0, Jm, O.FRAME_BODY if body else O.FRAME_WORLD), 6
    else:
        model, om, mctl = RD.Quadrotor(bodyframe=body), O.quadrotor(O.FRAME_BODY if body else O.FRAME_WORLD), 4
    N = 64 * 3 + 17
    Zn = H.rand_rigid(rng, N, mctl, 0.5 if mctl == 4 else 0.0, 1.5 if mctl == 4 else 1.0)
    t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
    dt[-1] = 0.0
    dt[70] = 0.0
    for rule, orule in ((RD.Euler, O.EULER), (RD.RK2, O.RK2), (RD.RK3, O.RK3), (RD.RK4, O.RK4)):
        dm = RD.DiscretizedDynamics(model, rule)
        Z = traj(model, Zn, t, dt, "cuda")
        J = RD.alloc_jacobians(dm, N, Z.data)
        Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
        RD.jacobian_(RD.StaticReturn(), RD.FiniteDifference(), dm, J, Y, Z)
        Jfd, Yo = om.jacobian(orule, O.FINITE_DIFF, Zn, t, dt)
        Jex, _ = om.jacobian(orule, O.FORWARD_AD, Zn, t, dt)
        Jg = host(J)
        np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
        # same differences up to the rounding of two evaluation orders (FMA contraction, sincos_, reciprocals) divided by eps = 2^-26 max(1,|x|):
        # a few ulps of |y| <= 4 (quadrotor: gravity in v) is 3e-7
        np.testing.assert_allclose(Jg, Jfd, rtol=0, atol=4e-7)
        np.testing.assert_allclose(Jg, Jex, rtol=0, atol=1e-6)
        # blocks a column cannot reach are exact zeros, like the full evaluation: w and q rows of the r, v columns; control columns of terminal knots
        assert np.all(Jg[:, 3:7, 0:3] == 0.0) and np.all(Jg[:, 10:13, 0:3] == 0.0) and np.all(Jg[:, 3:7, 7:10] == 0.0) and np.all(Jg[:, 10:13, 7:10] == 0.0)
        assert np.all(Jg[-1][:, 13:] == 0.0) and np.all(Jg[70][:, 13:] == 0.0)


@pytest.mark.parametrize("shape", [(64, 9), (130, 6), (37, 14), (200, 3), (64, 2), (5, 1), (256, 41)])
@pytest.mark.parametrize("variant", ["satellite_world_rk4", "satellite_body_rk3", "satellite_fullJ_rk2", "quadrotor_world_rk4", "quadrotor_body_euler",
                                     "satellite_world_rk4_f32"])
def test_cascade_rollout_shapes_layouts_and_rules(variant, shape):
    """The rigid bodies' rollout kernel (one wavefront per cascade block w | q | r,v plus an IO wavefront, csrc/cascade.hpp) against the oracle's
    sequential rollout: whole and ragged 64-trajectory groups, step counts that are / are not multiples of the IO chunk (4), one and two
    knots, knot-contiguous and time-major stacks, every explicit rule, world- and body-frame velocities, diagonal and full inertia, a
    terminal knot point with dt = 0 in the middle of the data (its control reads as zero), with and without a separate x0."""
    M, K = shape
    rng = np.random.default_rng(1000 + 7 * M + K)
    body = "_body_" in variant
    if variant.startswith("satellite"):
        Jm = np.array([[2.0, 0.1, 0.0], [0.1, 3.0, 0.2], [0.0, 0.2, 1.0]]) if "fullJ" in variant else np.diag([2.0, 3.0, 1.0])
        model, om, mctl = RD.Satellite(mass=2.0, J=Jm, bodyframe=body), O.satellite(2.0, Jm, O.FRAME_BODY if body else O.FRAME_WORLD), 6
    else:
        model, om, mctl = RD.Quadrotor(bodyframe=body), O.quadrotor(O.FRAME_BODY if body else O.FRAME_WORLD), 4
    rule, orule = {"rk4": (RD.RK4, O.RK4), "rk3": (RD.RK3, O.RK3), "rk2": (RD.RK2, O.RK2), "euler": (RD.Euler, O.EULER)}[variant.split("_")[2]]
    f32 = variant.endswith("_f32")
    dm = RD.DiscretizedDynamics(model, rule)
    data = H.rand_rigid(rng, M * K, mctl, 0.5 if mctl == 4 else 0.0, 1.5 if mctl == 4 else 1.0).reshape(M, K, 13 + mctl)
    t = np.broadcast_to(np.arange(K) * 0.02, (M, K)).copy()
    dt = np.full((M, K), 0.02)
    if K > 4:
        dt[::3, K // 2] = 0.0                              # terminal-style knots inside some trajectories: x_{k+1} = x_k + 0 * f, control ignored
    x0 = data[:, 0, :13].copy(); x0[:, 7:] *= 0.5
    for tm in (False, True):
        for with_x0 in (False, True):
            ref = om.rollout(orule, data, t, dt, X0=x0 if with_x0 else None)
            order = (1, 0, 2) if tm else (0, 1, 2)
            cast = (lambda a: a.astype(np.float32)) if f32 else (lambda a: a)
            stack = {"data": torch.as_tensor(cast(np.ascontiguousarray(data.transpose(order)))).cuda(),
                     "t": torch.as_tensor(np.ascontiguousarray(t.T if tm else t)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T if tm else dt)).cuda()}
            J = RD.alloc_jacobians(dm, K * M, stack["data"])
            RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, stack, torch.as_tensor(cast(x0)).cuda() if with_x0 else None, time_major=tm)
            got = host(stack["data"]).astype(np.float64)
            got = got.transpose(order) if tm else got
            if f32:
                np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
            else:
                np.testing.assert_allclose(got, ref, rtol=0, atol=TOL)


@pytest.mark.parametrize("case", ["satellite_f64_rk4_32806x10", "satellite_f32_rk4_32772x12", "cartpole_f64_rk4_32770x18", "quadrotor_f64_rk3_32768x9",
                                  "satellite_f64_euler_33000x1"])
def test_streaming_rollout_of_big_batches(case):
    """From two 64-trajectory groups per CU on (32 768 trajectories on an MI355X) rollout_ streams the stack through LDS in chunks of 4 / 8 knot points
    (csrc/rollout_stream.hpp: whole 16-byte units in, the states advanced in place, whole lines out -- the controls are written back unchanged).
    Against the oracle's sequential rollout of EVERY trajectory: both layouts, a ragged last group, step counts that are not multiples of the
    chunk, a single knot, a dt = 0 knot inside some trajectories, with and without a separate x0, fp32; shapes whose runs do not start on
    16-byte boundaries (odd K in the reference's layout) and models with small knot points (fp32 rigid bodies, the cartpole) take the
    one-lane-per-trajectory kernel and must give the same answer.  The controls, the times and the knot points behind a trajectory's end must
    come back untouched; implicit rules refuse time-major stacks."""
    parts = case.split("_")
    M, K = (int(v) for v in parts[3].split("x"))
    rng = np.random.default_rng(4242 + K)
    f32 = parts[1] == "f32"
    rule, orule = {"rk4": (RD.RK4, O.RK4), "rk3": (RD.RK3, O.RK3), "euler": (RD.Euler, O.EULER)}[parts[2]]
    if parts[0] == "cartpole":
        model, om, n, mctl = RD.Cartpole(), O.cartpole(), 4, 1
        data = rng.random((M, K, 5))
    else:
        mctl = 4 if parts[0] == "quadrotor" else 6
        model, om, n = (RD.Quadrotor(), O.quadrotor(), 13) if mctl == 4 else (RD.Satellite(), O.satellite(), 13)
        data = H.rand_rigid(rng, M * K, mctl, 0.5 if mctl == 4 else 0.0, 1.5 if mctl == 4 else 1.0).reshape(M, K, 13 + mctl)
    dm = RD.DiscretizedDynamics(model, rule)
    t = np.broadcast_to(np.arange(K) * 0.02, (M, K)).copy()
    dt = np.full((M, K), 0.02)
    if K > 4:
        dt[::5, K // 2] = 0.0
    x0 = data[:, 0, :n].copy(); x0[:, n // 2:] *= 0.5
    cast = (lambda a: a.astype(np.float32)) if f32 else (lambda a: a)
    for tm in (False, True):
        for with_x0 in (False, True):
            ref = om.rollout(orule, data, t, dt, X0=x0 if with_x0 else None)
            order = (1, 0, 2) if tm else (0, 1, 2)
            stack = {"data": torch.as_tensor(cast(np.ascontiguousarray(data.transpose(order)))).cuda(),
                     "t": torch.as_tensor(np.ascontiguousarray(t.T if tm else t)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T if tm else dt)).cuda()}
            t_before, dt_before = stack["t"].clone(), stack["dt"].clone()
            RD.rollout_(RD.StaticReturn(), dm, stack, torch.as_tensor(cast(x0)).cuda() if with_x0 else None, time_major=tm)
            got = host(stack["data"]).astype(np.float64)
            got = got.transpose(order) if tm else got
            if f32:
                if not np.allclose(got[:, :, :n], ref[:, :, :n], rtol=2e-4, atol=2e-4):
                    # (round 6: this comparison failed ONCE, under rocprofv3, and 1 600 launches of the same case did not reproduce it
                    # -- tools/repro_rollout_flake.py; say where it differs and whether a second read of the device memory differs too)
                    bad = ~np.isclose(got[:, :, :n], ref[:, :, :n], rtol=2e-4, atol=2e-4)
                    again = host(stack["data"]).astype(np.float64)
                    again = again.transpose(order) if tm else again
                    pytest.fail(f"time_major={tm} x0={with_x0}: {int(bad.sum())} values differ in trajectories {sorted(set(np.nonzero(bad)[0].tolist()))[:40]}, "
                                f"knots {sorted(set(np.nonzero(bad)[1].tolist()))}; exact zeros among them: {int((got[:, :, :n][bad] == 0).sum())}; a second read of the "
                                f"device memory differs from the oracle in {int((~np.isclose(again[:, :, :n], ref[:, :, :n], rtol=2e-4, atol=2e-4)).sum())} values; "
                                f"kernel {RD.last_kernel_name()}")
                np.testing.assert_array_equal(got[:, :, n:], cast(data[:, :, n:]).astype(np.float64))
            else:
                np.testing.assert_allclose(got[:, :, :n], ref[:, :, :n], rtol=0, atol=TOL)
                np.testing.assert_array_equal(got[:, :, n:], data[:, :, n:])           # the controls: bit for bit
            assert torch.equal(stack["t"], t_before) and torch.equal(stack["dt"], dt_before)
    with pytest.raises(Exception, match="trajectory-major"):
        RD.rollout_(RD.StaticReturn(), RD.DiscretizedDynamics(model, RD.ImplicitMidpoint), stack, time_major=True)
    # host arrays, time-major, a small batch: rdhip_rollout_layout through the staging path and the latency-regime kernels
    Ms = 70
    hst = {"data": np.ascontiguousarray(cast(data[:Ms]).transpose(1, 0, 2)), "t": np.ascontiguousarray(t[:Ms].T), "dt": np.ascontiguousarray(dt[:Ms].T)}
    RD.rollout_(RD.StaticReturn(), dm, hst, None, time_major=True)
    refs = om.rollout(orule, data[:Ms], t[:Ms], dt[:Ms])
    np.testing.assert_allclose(hst["data"].astype(np.float64).transpose(1, 0, 2)[:, :, :n], refs[:, :, :n], rtol=2e-4 if f32 else 0, atol=2e-4 if f32 else TOL)


@pytest.mark.parametrize("case", ["satellite_tm_large", "satellite_traj_major", "cartpole_tm_large", "quadrotor_tm_host", "custom_tm_large",
                                  "satellite_implicit_traj_major"])
def test_rollout_then_jacobians_in_one_call(case):
    """rdhip_rollout_jacobian = rollout! + jacobian! at every knot of the simulated trajectories, trajectory-major or time-major stacks:
    the result must be the oracle's rollout followed by the oracle's Jacobians, and later work on the caller's stream must see it."""
    rng = np.random.default_rng(31)
    tm = "_tm_" in case
    if case.startswith("satellite"):
        model, om, mctl = RD.Satellite(), O.satellite(), 6
    elif case.startswith("quadrotor"):
        model, om, mctl = RD.Quadrotor(), O.quadrotor(), 4
    elif case.startswith("custom"):
        model, om, mctl = RD.CustomModel(13, 6, dynamics=H.SATELLITE_SRC, params=[1.0, 1.0, 1.0, 1.0], liestate=RD.LieState(RD.QuatRotation, (3, 6))), O.satellite(), 6
    else:
        model, om, mctl = RD.Cartpole(), O.cartpole(), 1
    implicit = "implicit" in case
    Q, OQ = (RD.ImplicitMidpoint, None) if implicit else (RD.RK4, O.RK4)
    dm = RD.DiscretizedDynamics(model, Q)
    M, K = (9000, 11) if "large" in case else (70, 9)
    n = RD.state_dim(model)
    data = (H.rand_rigid(rng, M * K, mctl) if n == 13 else rng.random((M * K, 5))).reshape(M, K, n + mctl)
    t = np.broadcast_to(np.arange(K) * 0.05, (M, K)).copy()
    dt = np.full((M, K), 0.05); dt[:, -1] = 0.0
    x0 = data[:, 0, :n].copy() * 1.0
    if n == 13:
        x0[:, 7:] *= 0.5
    if implicit:
        dt[:, -1] = 0.05
        ref = data.copy(); ref[:, 0, :n] = x0
        for k in range(1, K):
            ref[:, k, :n] = om.implicit_midpoint(ref[:, k - 1], t[:, k - 1], dt[:, k - 1], want_jacobian=False)[0]
        Jref = om.implicit_midpoint(ref.reshape(M * K, -1), t.reshape(-1), dt.reshape(-1), want_jacobian=True)[1]
    else:
        ref = om.rollout(OQ, data, t, dt, X0=x0)
        Jref = om.jacobian(OQ, O.FORWARD_AD, ref.reshape(M * K, -1), t.reshape(-1), dt.reshape(-1))[0]
    order = (1, 0, 2) if tm else (0, 1, 2)
    dz, tz, dtz = np.ascontiguousarray(data.transpose(order)), np.ascontiguousarray(t.T if tm else t), np.ascontiguousarray(dt.T if tm else dt)
    on_host = case.endswith("_host")
    dev = (lambda a: a) if on_host else (lambda a: torch.as_tensor(a).cuda())
    stack = {"data": dev(dz), "t": dev(tz), "dt": dev(dtz)}
    J = RD.alloc_jacobians(dm, K * M, stack["data"])
    diff = RD.ImplicitFunctionTheorem(RD.ForwardAD()) if implicit else RD.ForwardAD()
    RD.rollout_jacobian_(RD.StaticReturn(), diff, dm, J, stack, dev(x0), time_major=tm)
    after = None if on_host else J.sum()                  # work queued on the caller's stream right after the call
    got = host(stack["data"])
    np.testing.assert_allclose(got.transpose(order) if tm else got, ref, rtol=0, atol=1e-9 if implicit else TOL)
    Jg = host(J).reshape((K, M) + Jref.shape[1:]).transpose(1, 0, 2, 3) if tm else host(J).reshape((M, K) + Jref.shape[1:])
    np.testing.assert_allclose(Jg.reshape(Jref.shape), Jref, rtol=0, atol=1e-8 if implicit else TOL)
    if after is not None:
        assert abs(float(after) - float(Jref.sum())) <= 1e-6 * max(1.0, abs(float(Jref.sum())))


@pytest.mark.parametrize("case", ["satellite_f64_rk4_512x41", "satellite_f64_rk4_130x14_ragged", "quadrotor_f64_rk3_192x6", "satellite_f32_rk4_132x30",
                                  "satellite_f64_rk4_64x1", "satellite_f64_rk4_66x2", "satellite_body_f64_euler_128x9", "satellite_f64_rk4_129x7_unaligned",
                                  "satellite_f32_rk4_130x5_unaligned", "quadrotor_f32_rk4_77x9_unaligned", "satellite_f64_rk3_1040x11"])
def test_fused_rollout_jacobian_overlaps_in_one_launch(case):
    """rdhip_rollout_jacobian on stacks of a rigid body is ONE launch (csrc/fused.hpp): rollout workgroups publish every finished chunk of 4 steps
    in a progress word, Jacobian workgroups on the other compute units consume the chunks behind them -- time-major stacks as 64-trajectory
    slabs per knot, trajectory-major stacks (the reference's layout) as gathered tiles of 4 trajectories x 16 knots.  It must reproduce the
    oracle's sequential rollout and Jacobians, agree with the two-call path (states bit for bit: rdhip_rollout runs the SAME kernel with rollout
    workgroups only; Jacobians to the last bit or two: the same column code in a separately compiled kernel), survive being called again
    (the words of an earlier call must not satisfy a later one) and being replayed from a captured graph; time-major stacks whose tiles do not
    start on 16-byte boundaries (odd M in fp64, M % 4 != 0 in fp32) take the two launches, trajectory-major stacks take any shape."""
    parts = case.split("_")
    body = parts[1] == "body"
    if body:
        parts.pop(1)
    name, prec, rname, shape = parts[0], parts[1], parts[2], parts[3]
    M, K = (int(v) for v in shape.split("x"))
    rng = np.random.default_rng(900 + M + K)
    if name == "satellite":
        model, om, mctl = RD.Satellite(bodyframe=body), O.satellite(1.0, np.eye(3), O.FRAME_BODY if body else O.FRAME_WORLD), 6
    else:
        model, om, mctl = RD.Quadrotor(), O.quadrotor(), 4
    rule, orule = {"rk4": (RD.RK4, O.RK4), "rk3": (RD.RK3, O.RK3), "euler": (RD.Euler, O.EULER)}[rname]
    f32 = prec == "f32"
    cast = (lambda a: a.astype(np.float32)) if f32 else (lambda a: a)
    dm = RD.DiscretizedDynamics(model, rule)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    for trial in range(4):                                  # calls on the same context: every call has its own epoch; both layouts
        tm = trial % 2 == 0
        expect_fused = not (tm and "unaligned" in case)
        data = H.rand_rigid(rng, M * K, mctl, 0.5 if mctl == 4 else 0.0, 1.5 if mctl == 4 else 1.0).reshape(M, K, 13 + mctl)
        t = np.broadcast_to(np.arange(K) * 0.02, (M, K)).copy()
        dt = np.full((M, K), 0.02); dt[:, -1] = 0.0
        x0 = data[:, 0, :13].copy(); x0[:, 7:] *= 0.5
        with_x0 = trial != 1
        ref = om.rollout(orule, data, t, dt, X0=x0 if with_x0 else None)
        Jref = om.jacobian(orule, O.FORWARD_AD, ref.reshape(M * K, -1), t.reshape(-1), dt.reshape(-1))[0].reshape(M, K, 13, 13 + mctl)
        order = (1, 0, 2) if tm else (0, 1, 2)
        mk = lambda: {"data": torch.as_tensor(cast(np.ascontiguousarray(data.transpose(order)))).cuda(),       # noqa: E731
                      "t": torch.as_tensor(np.ascontiguousarray(t.T if tm else t)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T if tm else dt)).cuda()}
        X0 = torch.as_tensor(cast(x0)).cuda() if with_x0 else None
        st1 = mk()
        J1 = RD.alloc_jacobians(dm, K * M, st1["data"]).fill_(7.0)
        launches = RD.rollout_jacobian_(sig, AD, dm, J1, st1, X0, time_major=tm)
        assert launches == (1 if expect_fused else 2), (launches, tm)
        got = host(st1["data"]).astype(np.float64).transpose(order)
        Jg = host(J1).astype(np.float64).reshape((K, M, 13, 13 + mctl) if tm else (M, K, 13, 13 + mctl))
        Jg = Jg.transpose(1, 0, 2, 3) if tm else Jg
        if f32:
            np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(Jg, Jref, rtol=2e-3, atol=2e-3)
        else:
            np.testing.assert_allclose(got, ref, rtol=0, atol=TOL)
            np.testing.assert_allclose(Jg, Jref, rtol=0, atol=TOL)
        # the two-call path: rdhip_rollout (trajectory-major stack) on a fresh copy of the inputs, then rdhip_jacobian on the simulated knots
        st2 = {"data": torch.as_tensor(cast(data)).cuda(), "t": torch.as_tensor(t).cuda(), "dt": torch.as_tensor(dt).cuda()}
        RD.rollout_(sig, dm, st2, X0)
        assert torch.equal(st1["data"], st2["data"].transpose(0, 1) if tm else st2["data"])
        J2 = RD.alloc_jacobians(dm, K * M, st1["data"])
        flat = RD.PackedTrajectory(13, mctl, st1["data"].reshape(K * M, 13 + mctl), st1["t"].reshape(-1), st1["dt"].reshape(-1))
        RD.jacobian_(sig, AD, dm, J2, None, flat)
        # the Jacobian workgroups run the flagship's column code compiled into the fused kernel: a separate instantiation, whose FMA contraction
        # may differ in the last bit (like the column-split kernel of small batches)
        np.testing.assert_allclose(host(J1), host(J2), rtol=0, atol=1e-5 if f32 else 1e-13)
    if "unaligned" in case or K < 3:
        return
    # replayed from a captured graph (the recorded epoch is reused: the capture carries a node that clears the progress words)
    from rdhip import _lib as L
    for tm in (True, False):
        order = (1, 0, 2) if tm else (0, 1, 2)
        s1 = torch.cuda.Stream()
        with torch.cuda.stream(s1):
            st = {"data": torch.as_tensor(cast(np.ascontiguousarray(data.transpose(order)))).cuda(),
                  "t": torch.as_tensor(np.ascontiguousarray(t.T if tm else t)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T if tm else dt)).cuda()}
            J = RD.alloc_jacobians(dm, K * M, st["data"])
            RD.rollout_jacobian_(sig, AD, dm, J, st, X0, time_major=tm)          # once outside the capture (kernel attributes)
            torch.cuda.synchronize()
            ctx = L.context_for(0, s1.cuda_stream)
            ctx.begin_capture()
            RD.rollout_jacobian_(sig, AD, dm, J, st, X0, time_major=tm)
            g = ctx.end_capture()
            for rep in range(2):
                fresh = H.rand_rigid(rng, M * K, mctl, 0.5 if mctl == 4 else 0.0, 1.5 if mctl == 4 else 1.0).reshape(M, K, 13 + mctl)
                st["data"].copy_(torch.as_tensor(cast(np.ascontiguousarray(fresh.transpose(order)))))
                J.zero_()
                ctx.launch(g)
                torch.cuda.synchronize()
                ref = om.rollout(orule, fresh, t, dt, X0=x0 if with_x0 else None)
                Jr = om.jacobian(orule, O.FORWARD_AD, ref.reshape(M * K, -1), t.reshape(-1), dt.reshape(-1))[0].reshape(M, K, 13, 13 + mctl)
                np.testing.assert_allclose(host(st["data"]).astype(np.float64).transpose(order), ref, rtol=2e-4 if f32 else 0, atol=2e-4 if f32 else TOL)
                Jg = host(J).astype(np.float64).reshape((K, M, 13, 13 + mctl) if tm else (M, K, 13, 13 + mctl))
                np.testing.assert_allclose(Jg.transpose(1, 0, 2, 3) if tm else Jg, Jr, rtol=2e-3 if f32 else 0, atol=2e-3 if f32 else TOL)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_mixed_model_rollouts_and_jacobians_in_one_call(prec):
    """rdhip_rollout_jacobian_multi (BASELINE config 5's shape: a Monte-Carlo batch whose trajectories belong to different models): Cartpole,
    Satellite and Quadrotor stacks in ONE call, every model on its own internal stream between a fork and a join.  Each model's states and
    Jacobians must be what its own rdhip_rollout_jacobian call gives (states exactly, Jacobians to rounding), match the oracle, and be complete for work
    queued on the caller's stream right after the call; repeated calls must not see each other's progress words."""
    rng = np.random.default_rng(77)
    f32 = prec == "f32"
    cast = (lambda a: a.astype(np.float32)) if f32 else (lambda a: a)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    specs = [(RD.Cartpole(), O.cartpole(), 1, 132, 30), (RD.Satellite(), O.satellite(), 6, 192, 25), (RD.Quadrotor(), O.quadrotor(), 4, 68, 11)]
    for trial in range(2):
        funs, stacks, Js, x0s, refs = [], [], [], [], []
        for model, om, mctl, M, K in specs:
            n = RD.state_dim(model)
            data = (H.rand_rigid(rng, M * K, mctl, 0.5 if mctl == 4 else 0.0, 1.5 if mctl == 4 else 1.0) if n == 13 else rng.random((M * K, n + mctl))).reshape(M, K, n + mctl)
            t = np.broadcast_to(np.arange(K) * 0.02, (M, K)).copy(); dt = np.full((M, K), 0.02); dt[:, -1] = 0.0
            x0 = data[:, 0, :n].copy() * 0.9
            dm = RD.DiscretizedDynamics(model, RD.RK4)
            mk = lambda: {"data": torch.as_tensor(cast(np.ascontiguousarray(data.transpose(1, 0, 2)))).cuda(),       # noqa: E731
                          "t": torch.as_tensor(np.ascontiguousarray(t.T)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T)).cuda()}
            st = mk()
            funs.append(dm); stacks.append(st); Js.append(RD.alloc_jacobians(dm, K * M, st["data"]).fill_(3.0)); x0s.append(torch.as_tensor(cast(x0)).cuda())
            ref = om.rollout(O.RK4, data, t, dt, X0=x0)
            Jref = om.jacobian(O.RK4, O.FORWARD_AD, ref.reshape(M * K, -1), t.reshape(-1), dt.reshape(-1))[0].reshape(M, K, n, n + mctl)
            # the single-model call on a copy
            st1 = mk(); J1 = RD.alloc_jacobians(dm, K * M, st1["data"])
            RD.rollout_jacobian_(sig, AD, dm, J1, st1, x0s[-1], time_major=True)
            refs.append((ref, Jref, st1, J1, M, K, n, mctl))
        RD.rollout_jacobian_multi_(sig, AD, funs, Js, stacks, x0s, time_major=True)
        sums = [float(J.double().sum()) for J in Js]          # work on the caller's stream behind the call
        for p, (ref, Jref, st1, J1, M, K, n, mctl) in enumerate(refs):
            # states bit for bit (the same rollout kernel either way); Jacobians to the last bit or two: ONE model of a call takes the overlapped single
            # launch, the others run the same column code as a separate Jacobian kernel behind their rollout (rdhip.h, rdhip_rollout_jacobian_multi)
            assert torch.equal(stacks[p]["data"], st1["data"])
            assert (Js[p] - J1).abs().max().item() <= (2e-6 if f32 else 1e-14) * max(1.0, J1.abs().max().item())
            got = host(stacks[p]["data"]).astype(np.float64).transpose(1, 0, 2)
            Jg = host(Js[p]).astype(np.float64).reshape(K, M, n, n + mctl).transpose(1, 0, 2, 3)
            np.testing.assert_allclose(got, ref, rtol=3e-4 if f32 else 0, atol=3e-4 if f32 else TOL)
            np.testing.assert_allclose(Jg, Jref, rtol=3e-3 if f32 else 0, atol=3e-3 if f32 else TOL)
            assert abs(sums[p] - float(J1.double().sum())) <= 1e-9 * max(1.0, abs(sums[p]))


@pytest.mark.parametrize("variant", ["satellite_quat", "quadrotor_quat", "satellite_mrp", "quadrotor_rp", "satellite_quat_f32", "quadrotor_quat_f32", "satellite_mrp_f32",
                                     "quadrotor_rp_f32"])
def test_wrenches_and_wrench_jacobian(variant):
    """wrenches(model, x, u, t) = [forces; moments] and d wrenches / d z (src/rigidbody.jl:244-248; wrench_jacobian!, :315-320) for every knot point
    of a batch against the oracle's forward duals (which tests/test_oracle_kat.py holds to test/rigidbody_test.jl:26-46's closed form): ragged
    batches, one knot point, host arrays through the staging path, fp32, MRP / RodriguesParam attitudes; wrench_sparsity as in the reference's
    docstring example; models without a wrench refuse."""
    kind, rtag = variant.split("_")[:2]
    f32 = variant.endswith("_f32")
    R, rot = {"quat": (RD.QuatRotation, O.ROT_QUAT), "mrp": (RD.MRP, O.ROT_MRP), "rp": (RD.RodriguesParam, O.ROT_RP)}[rtag]
    rng = np.random.default_rng(71)
    if kind == "satellite":
        model, om, mctl = (RD.Body(R, False), O.satellite(2.0, np.diag([2.0, 3.0, 1.0]), O.FRAME_WORLD, rot), 6)
    else:
        model, om, mctl = (RD.Quadrotor(R), O.quadrotor(O.FRAME_WORLD, rot), 4)
    n = RD.state_dim(model)
    for N in (130, 1, 64):
        if rot == O.ROT_QUAT:
            Zn = H.rand_rigid(rng, N, mctl, 0.5 if mctl == 4 else 0.0, 1.5 if mctl == 4 else 1.0)
        else:
            Zn = np.concatenate([rng.random((N, 3)), 0.6 * rng.random((N, 3)) - 0.3, rng.random((N, 6)), 0.5 + rng.random((N, mctl))], axis=1)
        t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
        Wo, Fo = om.wrenches(Zn, t), om.wrench_jacobian(Zn, t)
        for where in ("cuda", None):
            Z = traj(model, Zn, t, dt, where, dtype=np.float32 if f32 else np.float64)
            W, F = RD.wrenches(model, Z), RD.wrench_jacobian(model, Z)
            assert tuple(W.shape) == (N, 6) and tuple(F.shape) == (N, 6, n + mctl)
            if f32:
                np.testing.assert_allclose(host(W).astype(np.float64), Wo, rtol=2e-5, atol=2e-5)
                np.testing.assert_allclose(host(F).astype(np.float64), Fo, rtol=2e-5, atol=2e-5)
            else:
                np.testing.assert_allclose(host(W), Wo, rtol=0, atol=TOL)
                np.testing.assert_allclose(host(F), Fo, rtol=0, atol=TOL)
    # one knot point, (x, u, t) form
    w1 = RD.wrenches(model, Zn[0, :n], Zn[0, n:], 0.3)
    np.testing.assert_allclose(np.asarray(w1), Wo[0], rtol=0, atol=1e-12)
    js = RD.wrench_sparsity(model)
    np.testing.assert_array_equal(js, np.array([[False, True, False, False, True], [False, False, False, False, True]]))     # src/rigidbody.jl:339-345
    assert np.all((np.abs(Fo).max(axis=0).reshape(6, -1)[:3, :3] == 0))                                                        # dF/dr = 0, as the pattern says
    with pytest.raises(Exception):
        RD.wrenches(RD.Cartpole(), traj(RD.Cartpole(), rng.random((4, 5)), np.zeros(4), np.full(4, 0.1), "cuda"))


@pytest.mark.parametrize("rtag", ["mrp", "rp"])
def test_three_parameter_rigid_bodies_rollout_implicit_midpoint_and_finite_difference(rtag):
    """RigidBody{MRP} / RigidBody{RodriguesParam} (src/rigidbody.jl:80-107, :213-236; test/rigidbody_test.jl:79-98) beyond the golden fixtures: both
    velocity frames and wrenches through rollout!, ImplicitMidpoint (Newton solve + implicit-function-theorem Jacobian), FiniteDifference and a
    terminal knot point, against the oracle; dimension queries bit-exact; fp32 against the fp64 oracle."""
    R, rot = (RD.MRP, O.ROT_MRP) if rtag == "mrp" else (RD.RodriguesParam, O.ROT_RP)
    rng = np.random.default_rng(61)
    for body in (False, True):
        frame = O.FRAME_BODY if body else O.FRAME_WORLD
        for model, om, mctl, ulo, uhi in ((RD.Body(R, body), O.satellite(2.0, np.diag([2.0, 3.0, 1.0]), frame, rot), 6, 0.0, 1.0),
                                          (RD.Quadrotor(R, bodyframe=body), O.quadrotor(frame, rot), 4, 0.6, 1.9)):
            assert RD.dims(model)[:2] == (12, mctl) and RD.errstate_dim(model) == 12
            N = 64 * 2 + 11
            Zn = np.concatenate([rng.random((N, 3)), 0.6 * rng.random((N, 3)) - 0.3, rng.random((N, 6)), ulo + (uhi - ulo) * rng.random((N, mctl))], axis=1)
            t, dt = np.arange(N) * 0.1, np.full(N, 0.1); dt[-1] = 0.0
            Z = traj(model, Zn, t, dt, "cuda")
            dm = RD.DiscretizedDynamics(model, RD.RK4)
            J = RD.alloc_jacobians(dm, N, Z.data)
            Y = torch.empty((N, 12), dtype=torch.float64, device="cuda")
            RD.jacobian_(RD.StaticReturn(), RD.FiniteDifference(), dm, J, Y, Z)
            Jex, Yo = om.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
            np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
            np.testing.assert_allclose(host(J), Jex, rtol=0, atol=1e-6 * max(1.0, np.abs(Jex).max()))
            assert np.all(host(J)[-1][:, 12:] == 0.0)                          # terminal knot point: zero control columns
            dmi = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
            RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dmi, J, Y, Z)
            Xo, Jo, _ = om.implicit_midpoint(Zn, t, dt)
            np.testing.assert_allclose(host(Y), Xo, rtol=0, atol=TOL)
            np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL * max(1.0, np.abs(Jo).max()))
            # rollout!
            M, K = 70, 9
            data = np.stack([np.roll(Zn[:M], k, axis=0) for k in range(K)], axis=1)
            tt = np.broadcast_to(np.arange(K) * 0.05, (M, K)).copy(); dd = np.full((M, K), 0.05)
            ref = om.rollout(O.RK4, data, tt, dd)
            stack = {"data": torch.as_tensor(np.ascontiguousarray(data)).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
            RD.rollout_(RD.StaticReturn(), dm, stack)
            np.testing.assert_allclose(host(stack["data"]), ref, rtol=0, atol=TOL)
            # fp32 against the fp64 oracle
            Z32 = traj(model, Zn, t, dt, "cuda", dtype=np.float32)
            J32 = RD.alloc_jacobians(dm, N, Z32.data)
            RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J32, None, Z32)
            Jo32, _ = om.jacobian(O.RK4, O.FORWARD_AD, Zn.astype(np.float32).astype(np.float64), t, dt)
            np.testing.assert_allclose(host(J32), Jo32, rtol=1e-4, atol=1e-4)


def test_custom_model_with_a_declared_cascade_gets_the_cascade_kernels():
    """rdhip_model_set_cascade: a custom rigid body (the satellite as a source string) that declares w | q | r, v runs FiniteDifference along the
    cascade, rollout! with one wavefront per block and the block-by-block Newton solve -- same answers as the oracle and as the undeclared model;
    upstream blocks of the FiniteDifference columns are exact zeros; a false declaration is refused with the offending entry named."""
    rng = np.random.default_rng(71)
    mk = lambda: RD.CustomModel(13, 6, H.SATELLITE_SRC, params=(2.0, 2.0, 3.0, 1.0), liestate=(RD.QuatRotation, (3, 6)))
    plain, model = mk(), mk()
    model.set_cascade([2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 0, 0, 0], [2, 2, 2, 0, 0, 0])
    om = O.body()
    N = 64 * 3 + 17
    Zn = H.rand_rigid(rng, N, 6)
    t, dt = np.arange(N) * 0.1, np.full(N, 0.1); dt[-1] = 0.0; dt[70] = 0.0
    Z = traj(model, Zn, t, dt, "cuda")
    for Q, rule in ((RD.RK4, O.RK4), (RD.RK2, O.RK2)):
        dm, dp = RD.DiscretizedDynamics(model, Q), RD.DiscretizedDynamics(plain, Q)
        J = RD.alloc_jacobians(dm, N, Z.data); Jp = RD.alloc_jacobians(dp, N, Z.data)
        Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
        RD.jacobian_(RD.StaticReturn(), RD.FiniteDifference(), dm, J, Y, Z)
        RD.jacobian_(RD.StaticReturn(), RD.FiniteDifference(), dp, Jp, None, Z)
        Jex, Yo = om.jacobian(rule, O.FORWARD_AD, Zn, t, dt)
        np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(J), Jex, rtol=0, atol=1e-6)
        np.testing.assert_allclose(host(J), host(Jp), rtol=0, atol=4e-7)          # two evaluation orders of the same differences, divided by eps
        Jg = host(J)
        assert np.all(Jg[:, 3:7, 0:3] == 0.0) and np.all(Jg[:, 10:13, 0:3] == 0.0) and np.all(Jg[:, 10:13, 3:7] == 0.0) and np.all(Jg[-1][:, 13:] == 0.0)
    # rollout!: the cascade kernel (64 x K chunks through the IO wavefront, a ragged group, a short trajectory)
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    for M, K in ((130, 9), (64, 26), (5, 3)):
        data = H.rand_rigid(rng, M * K, 6).reshape(M, K, 19)
        tt = np.broadcast_to(np.arange(K) * 0.05, (M, K)).copy(); dd = np.full((M, K), 0.05)
        ref = om.rollout(O.RK4, data, tt, dd)
        stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
        RD.rollout_(RD.StaticReturn(), dm, stack)
        np.testing.assert_allclose(host(stack["data"]), ref, rtol=0, atol=TOL)
    # ImplicitMidpoint: block-by-block Newton + the IFT tile kernel
    dmi = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
    J = RD.alloc_jacobians(dmi, N, Z.data)
    Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dmi, J, Y, Z)
    Xo, Jo, _ = om.implicit_midpoint(Zn, t, dt)
    np.testing.assert_allclose(host(Y), Xo, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(J), Jo, rtol=0, atol=TOL)
    # implicit rollout along the declared cascade (one wavefront per block solving its block per step), a ragged group and a full one
    for M, K in ((70, 12), (128, 9)):
        data = H.rand_rigid(rng, M * K, 6).reshape(M, K, 19)
        tt, dd = np.tile(np.arange(K) * 0.05, (M, 1)), np.full((M, K), 0.05)
        ref = data.copy()
        for k in range(1, K):
            ref[:, k, :13] = om.implicit_midpoint(ref[:, k - 1], tt[:, k - 1], dd[:, k - 1], want_jacobian=False)[0]
        stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
        RD.rollout_(RD.StaticReturn(), dmi, stack)
        np.testing.assert_allclose(host(stack["data"]), ref, rtol=0, atol=TOL)
    # a false promise: the attitude placed UPSTREAM of the angular velocity it depends on
    liar = mk()
    liar.set_cascade([2, 2, 2, 0, 0, 0, 0, 2, 2, 2, 1, 1, 1], [2, 2, 2, 1, 1, 1])
    with pytest.raises(ValueError, match="declared cascade is not one"):
        RD.rollout_(RD.StaticReturn(), RD.DiscretizedDynamics(liar, RD.RK4), {"data": torch.as_tensor(H.rand_rigid(rng, 64 * 5, 6).reshape(64, 5, 19)).cuda(),
                                                                                 "t": torch.zeros((64, 5), dtype=torch.float64, device="cuda"),
                                                                                 "dt": torch.full((64, 5), 0.1, dtype=torch.float64, device="cuda")})


def test_custom_rigid_body_from_forces_and_moments_source():
    """RD.CustomRigidBody: the reference's way of defining a rigid body -- `forces` and `moments` methods on a RigidBody subtype (src/rigidbody.jl:244-258;
    test/quadrotor.jl:56-96) -- with the two wrench functions given as HIP source.  The quadrotor written that way must reproduce the compiled-in
    quadrotor's oracle in both velocity frames for dynamics, every diff method, rollout!, ImplicitMidpoint and the error-state operators."""
    rng = np.random.default_rng(73)
    from examples.custom_models import QUADROTOR_FORCES_SRC as forces, QUADROTOR_MOMENTS_SRC as moments
    for body in (False, True):
        model = RD.CustomRigidBody(4, forces, moments, mass=0.5, J=np.diag([0.0023, 0.0023, 0.004]), params=(0.0, 0.0, -9.81, 0.1750, 1.0, 0.0245), bodyframe=body)
        om = O.quadrotor(O.FRAME_BODY if body else O.FRAME_WORLD)
        assert RD.dims(model)[:2] == (13, 4) and RD.errstate_dim(model) == 12
        N = 64 * 2 + 9
        Zn = H.rand_rigid(rng, N, 4, 0.6, 1.9); Zn[3, 13] = -0.2                      # one negative rotor command: max(0, kf u)
        t, dt = np.arange(N) * 0.1, np.full(N, 0.1); dt[-1] = 0.0
        Z = traj(model, Zn, t, dt, "cuda")
        np.testing.assert_allclose(host(RD.dynamics(model, Z)), om.dynamics(Zn, t, dt), rtol=0, atol=TOL)
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        J = RD.alloc_jacobians(dm, N, Z.data)
        Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
        Jo, Yo = om.jacobian(O.RK4, O.FORWARD_AD, Zn, t, dt)
        for diff, tol in ((RD.ForwardAD(), TOL), (RD.UserDefined(), TOL), (RD.FiniteDifference(), 1e-6)):
            RD.jacobian_(RD.StaticReturn(), diff, dm, J, Y, Z)
            np.testing.assert_allclose(host(J), Jo, rtol=0, atol=tol * max(1.0, np.abs(Jo).max()))
            np.testing.assert_allclose(host(Y), Yo, rtol=0, atol=TOL)
        M, K = 70, 11
        data = H.rand_rigid(rng, M * K, 4, 0.6, 1.9).reshape(M, K, 17)
        tt = np.broadcast_to(np.arange(K) * 0.05, (M, K)).copy(); dd = np.full((M, K), 0.05)
        stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
        RD.rollout_(RD.StaticReturn(), dm, stack)
        np.testing.assert_allclose(host(stack["data"]), om.rollout(O.RK4, data, tt, dd), rtol=0, atol=TOL)
        dmi = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
        RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dmi, J, Y, Z)
        Xo, Jio, _ = om.implicit_midpoint(Zn, t, dt)
        np.testing.assert_allclose(host(Y), Xo, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(J), Jio, rtol=0, atol=TOL * max(1.0, np.abs(Jio).max()))
        G = RD.alloc_errstate_jacobians(model, N, Z.data)
        RD.errstate_jacobian_(model, G, Z, clear=True)
        np.testing.assert_allclose(host(G), om.errstate_jacobian(Zn[:, :13]), rtol=0, atol=TOL)


def test_implicit_midpoint_fp32_against_the_fp64_oracle():
    """ImplicitMidpoint in fp32: the reference's default tolerance (1e-12) is below fp32 resolution -- its loop runs to the iteration limit and returns the
    root to rounding; the block iterations stop at a rounding-level floor instead (implicit.hpp block_tolerance) and must not fall back to the exact
    kernel for it.  Solve, IFT Jacobian and rollout against the fp64 oracle at rel 1e-4 (the fp32 bar of this suite)."""
    rng = np.random.default_rng(81)
    for model, om, mctl, ulo, uhi in ((RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0])), O.body(), 6, 0.0, 1.0), (RD.Quadrotor(), O.quadrotor(), 4, 0.6, 1.9)):
        N = 64 * 2 + 5
        Zn = H.rand_rigid(rng, N, mctl, ulo, uhi).astype(np.float32).astype(np.float64)
        t, dt = np.arange(N) * 0.05, np.full(N, 0.05)
        dmi = RD.DiscretizedDynamics(model, RD.ImplicitMidpoint)
        Z = traj(model, Zn, t, dt, "cuda", dtype=np.float32)
        J = RD.alloc_jacobians(dmi, N, Z.data)
        Y = torch.empty((N, 13), dtype=torch.float32, device="cuda")
        RD.jacobian_(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dmi, J, Y, Z)
        Xo, Jo, _ = om.implicit_midpoint(Zn, t, dt)
        np.testing.assert_allclose(host(Y), Xo, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(host(J), Jo, rtol=1e-4, atol=1e-4)
        M, K = 70, 12
        data = H.rand_rigid(rng, M * K, mctl, ulo, uhi).astype(np.float32).reshape(M, K, 13 + mctl)
        tt = np.broadcast_to(np.arange(K) * 0.05, (M, K)).copy(); dd = np.full((M, K), 0.05)
        stack = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
        RD.rollout_(RD.StaticReturn(), dmi, stack)
        got = host(stack["data"]).astype(np.float64)
        # every simulated knot satisfies the implicit equation of its predecessor to fp32 accuracy (checked with the fp64 oracle's solve of each step)
        for k in range(K - 1):
            Xk, _, _ = om.implicit_midpoint(got[:, k, :], tt[:, k], dd[:, k], want_jacobian=False)
            np.testing.assert_allclose(got[:, k + 1, :13], Xk, rtol=2e-4, atol=2e-4)


def test_flagship_at_a_million_knot_points_walks_xcd_tile_ranges():
    """From 64 tiles per workgroup on (1 M knot points on a whole MI355X) every XCD walks its own contiguous eighth of the tiles (kernels.hpp
    jac_split_loop); the smaller parity tests never reach that branch.  1 048 653 satellite knot points (a ragged last tile), outputs pre-filled with
    NaN: every element must be written, and every 997th knot point, the first two tiles and the last 70 knot points must match the oracle."""
    rng = np.random.default_rng(91)
    N = (1 << 20) + 77
    Zn = H.rand_rigid(rng, N, 6)
    t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
    model = RD.Satellite(); dm = RD.DiscretizedDynamics(model, RD.RK4)
    Z = traj(model, Zn, t, dt, "cuda")
    J = RD.alloc_jacobians(dm, N, Z.data); J.fill_(float("nan"))
    Y = torch.full((N, 13), float("nan"), dtype=torch.float64, device="cuda")
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, Y, Z)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(J).any()) and not bool(torch.isnan(Y).any())
    idx = np.unique(np.concatenate([np.arange(0, N, 997), np.arange(N - 70, N), np.arange(0, 130)]))
    Jo, Yo = O.satellite().jacobian(O.RK4, O.FORWARD_AD, Zn[idx], t[idx], dt[idx])
    ix = torch.as_tensor(idx).cuda()
    np.testing.assert_allclose(host(J[ix]), Jo, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(Y[ix]), Yo, rtol=0, atol=TOL)


def test_native_gather_really_calls_rccl_on_one_gpu():
    """The RCCL calls of csrc/comm.hip (dlopen of librccl, ncclGetUniqueId, ncclCommInitAll / ncclCommInitRank, ncclAllGather, grouped ncclSend /
    ncclRecv, ncclCommDestroy) cannot meet a second GPU on the test box; RDHIP_COMM_FORCE_RCCL=1 sends a 1-rank communicator through them anyway
    (the shard travels rank 0 -> rank 0).  Both ways of building the communicator -- one process driving the devices (a Julia / C host) and one
    process per GPU with an id (what bench.py --gpus N does) -- in a child process with a time limit: a broken RCCL binding must fail, not hang."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys
        os.environ["RDHIP_COMM_FORCE_RCCL"] = "1"
        sys.path[:0] = [%r, %r]
        import numpy as np, torch
        import rdhip as RD
        from rdhip import parallel as P
        x = torch.arange(1 << 17, dtype=torch.float64, device="cuda:0") * 0.5 + 3.0
        for make in (lambda: P.Comm.local([0]), lambda: P.Comm.from_torch_distributed(0)):
            comm = make()
            for mode in ("all", "root"):
                out = torch.full_like(x, -1.0)
                comm.gather(mode, [x], [x.numel() * 8], [out], root=0)
                comm.synchronize(); torch.cuda.synchronize()
                assert torch.equal(out, x), mode
            comm.close()
        print("RCCL_OK")
    """) % (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert "RCCL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_single_knot_point_api_reads_like_the_reference():
    # examples/cartpole.jl:41-57 restated with v0.4 calls
    model = RD.Cartpole()
    n, m = RD.dims(model)[:2]
    rng = np.random.default_rng(1)
    x, u = RD.rand(model, rng)
    z = RD.KnotPoint(x, u, 0.0, 0.1)
    om = O.cartpole()
    xdot = RD.dynamics(model, x, u)
    np.testing.assert_allclose(xdot, om.dynamics(z.z[None], 0.0, 0.1)[0], atol=TOL)
    D = RD.DynamicsJacobian(model)
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), model, D, np.zeros(n), z)
    np.testing.assert_allclose(D.data, om.jacobian(O.CONTINUOUS, O.FORWARD_AD, z.z[None], 0.0, 0.1)[0][0], atol=TOL)
    dmodel = RD.DiscretizedDynamics[RD.RK3](model)
    xn = RD.discrete_dynamics(dmodel, z)
    np.testing.assert_allclose(xn, om.discrete_dynamics(O.RK3, z.z[None], 0.0, 0.1)[0], atol=TOL)
    np.testing.assert_allclose(RD.discrete_dynamics(RD.RK3, model, z), xn, atol=0)            # v0.3 spelling
    np.testing.assert_allclose(RD.discrete_dynamics(dmodel, x, u, 0.0, 0.1), xn, atol=0)
    RD.discrete_jacobian_(RD.RK3, D, model, z)                                                  # v0.3 spelling
    np.testing.assert_allclose(D.data, om.jacobian(O.RK3, O.FORWARD_AD, z.z[None], 0.0, 0.1)[0][0], atol=TOL)
    assert D.A.shape == (4, 4) and D.B.shape == (4, 1)
    # errors mirror the reference
    with pytest.raises(NotImplementedError):
        RD.jacobian_(RD.StaticReturn(), RD.UserDefined(), RD.SatellitePair(), np.zeros((14, 20)), np.zeros(14), RD.KnotPoint(np.ones(14), np.ones(6), 0.0, 0.1))
    with pytest.raises(ValueError):
        RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dmodel, RD.alloc_jacobians(dmodel, 3), None, RD.PackedTrajectory(4, 1, np.zeros((2, 5)), 0.0, 0.1))
