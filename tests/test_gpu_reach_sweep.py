"""The dispatch lattice, swept: every kernel instantiation librdhip.so carries is launched here through the C ABI and checked against
the CPU oracle -- model kind x rotation x velocity frame x inertia structure x rule x diff method x dtype x entry point x size regime x alignment.
The reference has ONE generic method per operator (src/functionbase.jl:242-257) and every specialisation Julia compiles for it is one its tests can reach;
this file is what makes the same true of the library's kernels (tools/kernel_reach.sh counts what the suite launches against what the library holds).

How the big launches stay cheap on the CPU side: knot points are independent, so a batch that REPEATS a 203-knot base batch must return the
base batch's result repeated -- the oracle runs on 203 knots, the comparison of the whole output happens on the GPU.  The base batch has a ragged last
tile, a terminal knot point (dt = 0: zero control, src/knotpoint.jl:67) and varying time steps.  Size regimes (dispatch.hpp): 203 knots (four workgroups
per tile for the plans that have that form), 8 389 (the persistent tile loop), 98 375 (>= 8 tiles per workgroup: the dynamic tile schedule, fp64).

Tolerances: fp64 ForwardAD / UserDefined 1e-10 (x max(1, |J|)); FiniteDifference within 1e-6 of the exact Jacobian (test/integration_tests.jl:14); fp32 against
the fp64 oracle on the fp32-rounded inputs at rel 1e-4 (FiniteDifference: 2e-2 -- a forward difference with step sqrt(eps(Float32)) = 3.5e-4);
ImplicitMidpoint 1e-9 / 1e-8 (two Newton iterations stopped at tolerance 1e-12 differ by that times the conditioning: rdhip.h)."""
import json
import os
import zlib

import numpy as np
import pytest

import rdhip as RD
from oracle import rd_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASE = 203                      # 3 full tiles + 11 knots
MID = 8192 + 64 * 3 + 5         # 132 tiles: past the four-workgroups-per-tile form (at most 128 tiles on 256 CUs)
BIG = 98304 + 64 + 7            # 1 538 tiles: >= 8 per workgroup of the dynamic schedule's 192-workgroup grid
BIG_OF = {"liesat": 8 * 768 * 64 + 71}      # (its 45 KB tiles leave three workgroups per CU: the full grid of 768, 8 tiles each -- common.hpp dyn_grid)
RULES = [None, RD.Euler, RD.RK2, RD.RK3, RD.RK4]
ORULE = {None: O.CONTINUOUS, RD.Euler: O.EULER, RD.RK2: O.RK2, RD.RK3: O.RK3, RD.RK4: O.RK4}
SIG = RD.StaticReturn()
KERNELS = {}                    # case -> sorted kernel names it launched (written to gpurun_out/reach_sweep_kernels.json: what tools/kernel_reach.sh cross-checks)

FULLJ = np.array([[2.0, 0.1, -0.2], [0.1, 3.0, 0.3], [-0.2, 0.3, 1.5]])
QJ = np.array([[0.0023, 1e-4, -2e-4], [1e-4, 0.0023, 1.5e-4], [-2e-4, 1.5e-4, 0.004]])
A42 = np.array([[0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0], [-1.0, 0.3, -0.1, 0.0], [0.2, -2.0, 0.0, -0.2]])
B42 = np.array([[0.0, 0.0], [0.0, 0.0], [1.0, 0.0], [0.0, 0.5]])


def _lin(n, m):
    rng = np.random.default_rng(100 * n + m)
    return RD.LinearModel(rng.standard_normal((n, n)) * 0.4, rng.standard_normal((n, m)))


# name -> (factory, what the dispatch selects it by)
MODELS = {
    "cartpole": lambda: RD.Cartpole(),
    "di1": lambda: RD.DoubleIntegrator(1), "di2": lambda: RD.DoubleIntegrator(2), "di3": lambda: RD.DoubleIntegrator(3),
    "lin21": lambda: _lin(2, 1), "lin42": lambda: RD.LinearModel(A42, B42), "lin63": lambda: _lin(6, 3),
    "sat_wd": lambda: RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0])), "sat_wf": lambda: RD.Satellite(2.0, FULLJ),
    "sat_bd": lambda: RD.Satellite(2.0, np.diag([2.0, 3.0, 1.0]), bodyframe=True), "sat_bf": lambda: RD.Satellite(2.0, FULLJ, bodyframe=True),
    "quad_wd": lambda: RD.Quadrotor(), "quad_wf": lambda: RD.Quadrotor(J=QJ),
    "quad_bd": lambda: RD.Quadrotor(bodyframe=True), "quad_bf": lambda: RD.Quadrotor(J=QJ, gravity=(0.3, -0.2, -9.7), bodyframe=True),
    "sat_mrp_w": lambda: RD.Satellite(2.0, FULLJ, R=RD.MRP), "sat_rp_b": lambda: RD.Satellite(2.0, FULLJ, R=RD.RodriguesParam, bodyframe=True),
    "quad_mrp_b": lambda: RD.Quadrotor(R=RD.MRP, bodyframe=True), "quad_rp_w": lambda: RD.Quadrotor(R=RD.RodriguesParam),
    "pair": lambda: RD.SatellitePair(np.diag([1.0, 2.0, 1.5]), np.diag([2.0, 1.0, 3.0])), "liesat": lambda: RD.LieSatellite((1.0, 2.0, 1.5)),
}
DTYPES = {"f64": np.float64, "f32": np.float32}
CASES = [(m, d) for m in MODELS for d in DTYPES]
ROTCODE = {RD.QuatRotation: 0, RD.MRP: 1, RD.RodriguesParam: 2}


def oracle_of(model):
    """the oracle's twin: same kind, same parameter vector (oracle/rd_oracle.cpp keeps rdhip.h's enumerations), same rotation and velocity frame"""
    rot = ROTCODE[getattr(model, "rotation", RD.QuatRotation)]
    frame = 1 if getattr(model, "velocity_frame", "world") == "body" else 0
    return O.Model(model._kind, [float(v) for v in np.asarray(model._params(), dtype=np.float64)], rot=rot, frame=frame)


def base_inputs(name, model, N, seed):
    """rand(model)-like knot points (src/functionbase.jl:271, src/rigidbody.jl:50-59): unit quaternions, small three-parameter attitudes, rotor thrusts
    around trim for the quadrotor; times and time steps vary, knot 77 is terminal"""
    rng = np.random.default_rng(seed)
    n, m = RD.state_dim(model), RD.control_dim(model)
    Z = rng.random((N, n + m))

    def quat(N):
        q = rng.standard_normal((N, 4)); return q / np.linalg.norm(q, axis=1, keepdims=True)
    if name.startswith(("sat_", "quad_")):
        rot = ROTCODE[model.rotation]
        if rot == 0:
            Z[:, 3:7] = quat(N)
        else:
            Z[:, 3:6] = rng.uniform(-0.5, 0.5, (N, 3))
        if name.startswith("quad_"):
            Z[:, n:] = (9.81 * 0.5 / 4) * (0.5 + rng.random((N, m)))
    elif name == "pair":
        Z[:, 0:4] = quat(N); Z[:, 7:11] = quat(N)
    elif name == "liesat":
        Z[:, 3:7] = quat(N)
    elif name == "cartpole":
        Z[:, :4] -= 0.4
    t = np.cumsum(rng.uniform(0.01, 0.1, N))
    dt = rng.uniform(0.01, 0.06, N)
    if N > 77:
        dt[77] = 0.0
    return Z, t, dt


def tile_rows(a, N):
    reps = -(-N // a.shape[0])
    return np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:N])


class Case:
    """one (model, dtype): the base batch, its device copies at the three sizes, and the oracle's answers on the base batch (computed on demand)"""

    def __init__(self, name, dname):
        self.name, self.dname, self.T = name, dname, DTYPES[dname]
        self.model = MODELS[name]()
        self.om = oracle_of(self.model)
        self.n, self.m = RD.state_dim(self.model), RD.control_dim(self.model)
        Z, self.t, self.dt = base_inputs(name, self.model, BASE, seed=zlib.crc32(name.encode()) % 9973)
        self.Z = Z.astype(self.T).astype(np.float64)               # what the kernels see, exactly
        self.f32 = self.T == np.float32
        self._traj, self._oj, self.kernels = {}, {}, set()

    def traj(self, N):
        if N not in self._traj:
            self._traj[N] = RD.PackedTrajectory(self.n, self.m, torch.as_tensor(tile_rows(self.Z, N).astype(self.T)).cuda(),
                                                torch.as_tensor(tile_rows(self.t, N)).cuda(), torch.as_tensor(tile_rows(self.dt, N)).cuda())
        return self._traj[N]

    def fun(self, rule):
        return self.model if rule is None else RD.DiscretizedDynamics(self.model, rule)

    def oracle_jac(self, rule, diff):
        key = (rule, diff)
        if key not in self._oj:
            try:
                self._oj[key] = self.om.jacobian(ORULE[rule], diff, self.Z, self.t, self.dt)
            except NotImplementedError:
                self._oj[key] = None
        return self._oj[key]

    def note(self):
        k = RD.last_kernel_name()
        if k:
            self.kernels.add(k)
        return k

    def tol(self, diff_code, exp):
        s = max(1.0, float(np.abs(exp).max()))
        if diff_code == O.FINITE_DIFF:
            return (2e-2 if self.f32 else 1e-6) * s, 0.0
        return (1e-4 * s, 1e-4) if self.f32 else (1e-10 * s, 0.0)


_cases = {}


def case_of(name, dname):
    if (name, dname) not in _cases:
        _cases[(name, dname)] = Case(name, dname)
    return _cases[(name, dname)]


@pytest.fixture(scope="module", autouse=True)
def _dump_kernels():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "reach_sweep_kernels.json"), "w") as fh:
        json.dump({f"{k[0]}/{k[1]}": sorted(c.kernels) for k, c in _cases.items()}, fh, indent=1)


def jac_buffers(c, fun, N, aligned):
    """(J, Y) device outputs pre-filled with NaN; aligned=False: both start one element past a 16-byte boundary"""
    n, nz = c.n, c.n + c.m
    tdt = torch.float64 if not c.f32 else torch.float32
    if aligned:
        J = RD.alloc_jacobians(fun, N, c.traj(BASE).data)
        Y = torch.empty((N, n), dtype=tdt, device="cuda")
    else:
        J = torch.empty(N * n * nz + 1, dtype=tdt, device="cuda")[1:].view(N, nz, n).transpose(1, 2)
        Y = torch.empty(N * n + 1, dtype=tdt, device="cuda")[1:].view(N, n)
    J.fill_(float("nan")); Y.fill_(float("nan"))
    return J, Y


def assert_periodic(got, exp_base, atol, rtol, what):
    """got (N, ...) on the device must be exp_base (B, ...) repeated; compared on the device"""
    B, N = exp_base.shape[0], got.shape[0]
    e = torch.as_tensor(np.ascontiguousarray(exp_base)).cuda().double()
    g = got.double()
    full = N // B
    bound = atol + rtol * e.abs()
    if full:
        err = (g[:full * B].reshape((full, B) + tuple(g.shape[1:])) - e).abs().amax(dim=0)
        bad = ~(err <= bound)
        assert not bool(bad.any()), f"{what}: max error {float(err[~torch.isnan(err)].max()) if bool((~torch.isnan(err)).any()) else float('nan'):.3e} (NaN: {bool(torch.isnan(err).any())}), bound {atol:.1e} + {rtol:.1e} |x|"
    r = N - full * B
    if r:
        err = (g[full * B:] - e[:r]).abs()
        assert bool((err <= bound[:r]).all()), f"{what} (ragged end): max error {float(err.max()):.3e}"


# ------------------------------------------------------------------------------------------------------------------------------------------
# dynamics / discrete_dynamics: every rule (eval_kernel), and the implicit rule's Newton kernels
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,dname", CASES)
def test_sweep_evaluate(name, dname):
    c = case_of(name, dname)
    for rule in RULES:
        exp = c.om.dynamics(c.Z, c.t, c.dt) if rule is None else c.om.discrete_dynamics(ORULE[rule], c.Z, c.t, c.dt)
        for N in (BASE, MID):
            got = RD.evaluate(c.fun(rule), c.traj(N))
            c.note()
            atol, rtol = c.tol(O.FORWARD_AD, exp)
            assert_periodic(got, exp, atol, rtol, f"{name}/{dname} evaluate rule {rule.__name__ if rule else 'continuous'} N={N}")


# ------------------------------------------------------------------------------------------------------------------------------------------
# jacobian!: rule x diff x size regime x alignment
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,dname", CASES)
def test_sweep_jacobian(name, dname):
    c = case_of(name, dname)
    diffs = ((RD.ForwardAD(), O.FORWARD_AD), (RD.FiniteDifference(), O.FINITE_DIFF), (RD.UserDefined(), O.USER))
    for rule in RULES:
        fun = c.fun(rule)
        exact = c.oracle_jac(rule, O.FORWARD_AD)
        for diff, dcode in diffs:
            ref = c.oracle_jac(rule, dcode)
            # which (N, aligned) combinations select different kernels: the small form and the tile loop (ForwardAD), the unaligned fall-backs, the dynamic schedule
            combos = [(BASE, True), (BASE, False)]
            if dcode == O.FORWARD_AD:
                combos += [(MID, True), (MID, False)]
            if not c.f32:
                combos += [(BIG_OF.get(name, BIG), True)]
            for N, aligned in combos:
                J, Y = jac_buffers(c, fun, N, aligned)
                what = f"{name}/{dname} jacobian rule {rule.__name__ if rule else 'continuous'} {type(diff).__name__} N={N} aligned={aligned}"
                if ref is None:                                    # no user-defined Jacobian for this model: NotImplementedError on both sides (src/functionbase.jl:257)
                    with pytest.raises(NotImplementedError):
                        RD.jacobian_(SIG, diff, fun, J, Y, c.traj(N))
                    break
                RD.jacobian_(SIG, diff, fun, J, Y, c.traj(N))
                c.note()
                Jo = exact[0] if dcode == O.FINITE_DIFF else ref[0]    # FiniteDifference is held to the EXACT Jacobian, as the reference's tests do
                atol, rtol = c.tol(dcode, Jo)
                assert_periodic(J, Jo, atol, rtol, what)
                atol, rtol = c.tol(O.FORWARD_AD, ref[1])
                assert_periodic(Y, ref[1], atol, rtol, what + " f(z)")


# ------------------------------------------------------------------------------------------------------------------------------------------
# rdhip_jacobian_multi: inline table / device table (uniform and ragged) / long batches with the dynamic schedule; the batch-by-batch fallbacks
# ------------------------------------------------------------------------------------------------------------------------------------------
def _multi(c, fun, diff, sizes, aligned=True):
    trajs, Js, Ys = [], [], []
    for N in sizes:
        tr = c.traj(BASE) if N == BASE else RD.PackedTrajectory(c.n, c.m, c.traj(N).data.clone(), c.traj(N).t, c.traj(N).dt)
        J, Y = jac_buffers(c, fun, N, aligned)
        trajs.append(tr); Js.append(J); Ys.append(Y)
    launches = RD.jacobian_multi_(SIG, diff, fun, Js, Ys, trajs)
    return launches, Js, Ys


@pytest.mark.parametrize("name,dname", CASES)
def test_sweep_jacobian_multi(name, dname):
    c = case_of(name, dname)
    for rule in RULES:
        fun = c.fun(rule)
        ref = c.oracle_jac(rule, O.FORWARD_AD)
        atol, rtol = c.tol(O.FORWARD_AD, ref[0])
        shapes = {"inline": [BASE, 64, 130], "table uniform": [70] * 40, "table ragged": [70 + (i % 3) for i in range(40)]}
        if not c.f32:
            shapes["long"] = [BIG_OF.get(name, 98816) // 4] * 4    # 4 x 386 tiles: the dynamic tile schedule (>= 8 tiles per workgroup, >= 16 per batch)
        for tag, sizes in shapes.items():
            launches, Js, Ys = _multi(c, fun, RD.ForwardAD(), sizes)
            c.note()
            assert launches == 1, f"{name}/{dname} multi {tag}: {launches} launches"
            for J, Y in zip(Js, Ys):
                assert_periodic(J, ref[0], atol, rtol, f"{name}/{dname} multi {tag} rule {rule.__name__ if rule else 'continuous'}")
                assert_periodic(Y, ref[1], atol, rtol, f"{name}/{dname} multi {tag} f(z)")
    # what has no one-launch form runs batch by batch with the same results: unaligned outputs, FiniteDifference
    fun = c.fun(RD.RK4)
    ref, refd = c.oracle_jac(RD.RK4, O.FORWARD_AD), c.oracle_jac(RD.RK4, O.FINITE_DIFF)
    launches, Js, _ = _multi(c, fun, RD.ForwardAD(), [BASE, 64, 130], aligned=False)
    small = c.n + c.m <= 10 and name != "liesat"
    assert launches == (1 if small else 3)                         # (the small-model kernel takes any alignment)
    for J in Js:
        assert_periodic(J, ref[0], *c.tol(O.FORWARD_AD, ref[0]), f"{name}/{dname} multi unaligned")
    launches, Js, _ = _multi(c, fun, RD.FiniteDifference(), [BASE, 64, 130])
    assert launches == 3
    for J in Js:
        assert_periodic(J, ref[0], *c.tol(O.FINITE_DIFF, ref[0]), f"{name}/{dname} multi FiniteDifference")
    assert refd is not None
    # ... and from four batches on as gather + ONE rdhip_jacobian over the concatenation + scatter (api.hip multi_staged): three launches whatever the count
    six = [BASE, 64, 130, 70, 71, 203]
    launches, Js, Ys = _multi(c, fun, RD.FiniteDifference(), six)
    assert launches == 3, launches
    for J, Y in zip(Js, Ys):
        assert_periodic(J, ref[0], *c.tol(O.FINITE_DIFF, ref[0]), f"{name}/{dname} multi FiniteDifference, six batches")
        assert_periodic(Y, ref[1], *c.tol(O.FORWARD_AD, ref[1]), f"{name}/{dname} multi FiniteDifference, six batches, f(z)")
    launches, Js, Ys = _multi(c, fun, RD.ForwardAD(), six, aligned=False)
    assert launches == (1 if small else 3), launches
    for J, Y in zip(Js, Ys):
        assert_periodic(J, ref[0], *c.tol(O.FORWARD_AD, ref[0]), f"{name}/{dname} multi unaligned, six batches")
        assert_periodic(Y, ref[1], *c.tol(O.FORWARD_AD, ref[1]), f"{name}/{dname} multi unaligned, six batches, f(z)")


# ------------------------------------------------------------------------------------------------------------------------------------------
# error-state expansion, dynamics_error, dynamics_error_jacobian!
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,dname", CASES)
def test_sweep_error_expansion_and_dynamics_error(name, dname):
    c = case_of(name, dname)
    n, m, e = c.n, c.m, RD.errstate_dim(c.model)
    tdt = torch.float32 if c.f32 else torch.float64
    for rule in RULES[1:]:
        fun = c.fun(rule)
        # consecutive pairs of a batch that repeats the base batch: pair k of the repetition = pair k of [base; base[0]] -- one more row for the oracle
        Zx, tx, dtx = (np.concatenate([a, a[:1]]) for a in (c.Z, c.t, c.dt))
        Eo, Yo = c.om.error_expansion(ORULE[rule], Zx, tx, dtx)
        for N in (BASE, MID) + (() if c.f32 else (BIG_OF.get(name, BIG),)):
            E = RD.alloc_error_expansions(fun, N, c.traj(BASE).data); E.fill_(float("nan"))
            Y = torch.full((N - 1, n), float("nan"), dtype=tdt, device="cuda")
            RD.error_expansion_(fun, E, Y, c.traj(N))
            c.note()
            atol, rtol = c.tol(O.FORWARD_AD, Eo)
            assert_periodic(E, Eo, atol * (10 if not c.f32 else 1), rtol, f"{name}/{dname} error_expansion {rule.__name__} N={N}")
            assert_periodic(Y, Yo, *c.tol(O.FORWARD_AD, Yo), f"{name}/{dname} error_expansion f(z) {rule.__name__} N={N}")
        eo = c.om.dynamics_error(ORULE[rule], Zx, tx, dtx)
        got = RD.dynamics_error(fun, c.traj(MID)); c.note()
        assert_periodic(got, eo, *c.tol(O.FORWARD_AD, eo), f"{name}/{dname} dynamics_error {rule.__name__}")
        J2o, J1o, _ = c.om.dynamics_error_jacobian(ORULE[rule], O.FORWARD_AD, Zx, tx, dtx)
        J2, J1 = RD.alloc_jacobians(fun, MID - 1, c.traj(BASE).data), RD.alloc_jacobians(fun, MID - 1, c.traj(BASE).data)
        y2 = torch.empty((MID - 1, n), dtype=tdt, device="cuda")
        RD.dynamics_error_jacobian_(SIG, RD.ForwardAD(), fun, J2, J1, y2, None, c.traj(MID)); c.note()
        assert_periodic(J1, J1o, *c.tol(O.FORWARD_AD, J1o), f"{name}/{dname} dynamics_error_jacobian J1 {rule.__name__}")
        assert_periodic(J2, J2o, 0.0, 0.0, f"{name}/{dname} dynamics_error_jacobian J2 {rule.__name__}")
        assert_periodic(y2, eo, *c.tol(O.FORWARD_AD, eo), f"{name}/{dname} dynamics_error_jacobian y2 {rule.__name__}")
    assert e <= n


# ------------------------------------------------------------------------------------------------------------------------------------------
# rollout! and rollout! + jacobian!: rule x layout x batch-size regime
# ------------------------------------------------------------------------------------------------------------------------------------------
def _stack(c, M, K, tm, base_M=70, seed=5):
    """M trajectories of K knots repeating base_M base trajectories; returns (device stack, base data (base_M, K, nz), t, dt (base_M, K))"""
    nz = c.n + c.m
    Zb, _, _ = base_inputs(c.name, c.model, base_M * K, seed=seed + K)
    Zb = Zb.astype(c.T).astype(np.float64).reshape(base_M, K, nz)
    rng = np.random.default_rng(K)
    dtb = np.tile(rng.uniform(0.01, 0.04, K), (base_M, 1)); tb = np.cumsum(dtb, axis=1) - dtb
    data, tt, dd = tile_rows(Zb, M), tile_rows(tb, M), tile_rows(dtb, M)
    if tm:
        data, tt, dd = (np.ascontiguousarray(np.swapaxes(a, 0, 1)) for a in (data, tt, dd))
    st = {"data": torch.as_tensor(data.astype(c.T)).cuda(), "t": torch.as_tensor(tt).cuda(), "dt": torch.as_tensor(dd).cuda()}
    return st, Zb, tb, dtb


@pytest.mark.parametrize("name,dname", CASES)
def test_sweep_rollout_and_rollout_jacobian(name, dname):
    c = case_of(name, dname)
    nz = c.n + c.m
    for rule in RULES[1:]:
        fun = c.fun(rule)
        # (M, K): a latency-sized stack; 8 261 trajectories (more than half a chip of 64-trajectory groups); 32 808 (two groups per CU: the streaming regime,
        # chunks of 16-byte units), 32 809 time-major (an odd count: the tiles of a knot do not start on 16-byte boundaries)
        for M, K, layouts in ((70, 9, (False, True)), (8261, 3, (False, True)), (32808, 4, (False, True)), (32809, 3, (True,))):
            for tm in layouts:
                st, Zb, tb, dtb = _stack(c, M, K, tm)
                Zr = c.om.rollout(ORULE[rule], Zb, tb, dtb)
                RD.rollout_(SIG, fun, st, time_major=tm); c.note()
                got = st["data"].transpose(0, 1) if tm else st["data"]
                s = max(1.0, float(np.abs(Zr).max()))
                atol, rtol = ((2e-4 * s, 2e-4) if c.f32 else (1e-10 * s, 0.0))     # (fp32: K - 1 steps of rounding)
                assert_periodic(got, Zr, atol, rtol, f"{name}/{dname} rollout {rule.__name__} M={M} K={K} time_major={tm}")
        # the same rollouts on split arrays (rdhip_rollout_split: X written, U read): one streaming kernel for both layouts and every size
        for M, K, tm in ((70, 9, False), (70, 9, True), (333, 6, True)):
            st, Zb, tb, dtb = _stack(c, M, K, tm)
            Zr = c.om.rollout(ORULE[rule], Zb, tb, dtb)
            X = torch.full(tuple(st["data"].shape[:2]) + (c.n,), float("nan"), dtype=st["data"].dtype, device="cuda")
            if tm:
                X[0] = st["data"][0, :, :c.n]
            else:
                X[:, 0] = st["data"][:, 0, :c.n]
            U = (st["data"][:K - 1, :, c.n:] if tm else st["data"][:, :K - 1, c.n:]).contiguous()
            RD.rollout_split_(SIG, fun, X, U, st["t"], st["dt"], time_major=tm); c.note()
            got = X.transpose(0, 1) if tm else X
            s = max(1.0, float(np.abs(Zr).max()))
            atol, rtol = ((2e-4 * s, 2e-4) if c.f32 else (1e-10 * s, 0.0))
            assert_periodic(got, Zr[:, :, :c.n], atol, rtol, f"{name}/{dname} rollout_split {rule.__name__} M={M} K={K} time_major={tm}")
        # an initial state of its own (x0) and the Jacobians along the rollout in the same call
        shapes = [(72, 9, False), (72, 9, True)]                      # (a multiple of 4: the fp32 time-major one-launch form wants 16-byte aligned rows of tiles)
        if c.f32 and rule in (RD.RK3, RD.RK4) and name.startswith(("sat_", "quad_")) and ROTCODE[c.model.rotation] == 0:
            shapes += [(4096, 5, True), (4096, 20, False)]            # 64 rollout groups, bound by the Jacobian workgroups: the two-workgroups-per-CU forms (fused.hpp)
        for M, K, tm in shapes:
            st, Zb, tb, dtb = _stack(c, M, K, tm, seed=11)
            x0 = tile_rows(Zb[:, 0, :c.n] * 0.5 + 0.25 * Zb[:, 1, :c.n], M)
            if name.startswith(("sat_", "quad_")) and ROTCODE[c.model.rotation] == 0:
                x0[:, 3:7] /= np.linalg.norm(x0[:, 3:7], axis=1, keepdims=True)
            x0 = x0.astype(c.T).astype(np.float64)
            Zr = c.om.rollout(ORULE[rule], Zb, tb, dtb, X0=x0[:Zb.shape[0]])
            Jo, _ = c.om.jacobian(ORULE[rule], O.FORWARD_AD, Zr.reshape(-1, nz), tb.reshape(-1), dtb.reshape(-1))
            Jo = Jo.reshape(Zb.shape[0], K, c.n, nz)
            J = RD.alloc_jacobians(fun, M * K, st["data"]); J.fill_(float("nan"))
            launches = RD.rollout_jacobian_(SIG, RD.ForwardAD(), fun, J, st, x0=torch.as_tensor(x0.astype(c.T)).cuda(), time_major=tm); c.note()
            assert launches >= 1
            got = st["data"].transpose(0, 1) if tm else st["data"]
            s = max(1.0, float(np.abs(Zr).max()))
            atol, rtol = ((2e-4 * s, 2e-4) if c.f32 else (1e-10 * s, 0.0))
            assert_periodic(got, Zr, atol, rtol, f"{name}/{dname} rollout_jacobian states {rule.__name__} time_major={tm}")
            Jg = J.reshape(K, M, c.n, nz).transpose(0, 1) if tm else J.reshape(M, K, c.n, nz)
            sj = max(1.0, float(np.abs(Jo).max()))
            atol, rtol = ((1e-3 * sj, 1e-3) if c.f32 else (1e-10 * sj, 0.0))       # (fp32: the Jacobians are taken at the fp32 rollout's states)
            assert_periodic(Jg, Jo, atol, rtol, f"{name}/{dname} rollout_jacobian Jacobians {rule.__name__} time_major={tm}")


# ------------------------------------------------------------------------------------------------------------------------------------------
# DiscretizedDynamics{ImplicitMidpoint}: Newton solve, implicit-function-theorem Jacobian (aligned / unaligned), dynamics error, rollout
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,dname", CASES)
def test_sweep_implicit_midpoint(name, dname):
    c = case_of(name, dname)
    fun = RD.DiscretizedDynamics(c.model, RD.ImplicitMidpoint)
    ift = RD.ImplicitFunctionTheorem(RD.ForwardAD())
    Xo, Jo, _ = c.om.implicit_midpoint(c.Z, c.t, c.dt)
    sx, sj = max(1.0, float(np.abs(Xo).max())), max(1.0, float(np.abs(Jo).max()))
    tx, tj = ((1e-4 * sx, 1e-4), (2e-3 * sj, 2e-3)) if c.f32 else ((1e-9 * sx, 0.0), (1e-8 * sj, 0.0))
    for N in (BASE, MID):
        got = RD.evaluate(fun, c.traj(N)); c.note()
        assert_periodic(got, Xo, *tx, f"{name}/{dname} implicit discrete_dynamics N={N}")
        for aligned in (True, False):
            J, Y = jac_buffers(c, fun, N, aligned)
            RD.jacobian_(SIG, ift, fun, J, Y, c.traj(N)); c.note()
            assert_periodic(Y, Xo, *tx, f"{name}/{dname} implicit x_next N={N} aligned={aligned}")
            assert_periodic(J, Jo, *tj, f"{name}/{dname} implicit IFT Jacobian N={N} aligned={aligned}")
        J, _ = jac_buffers(c, fun, N, True)
        RD.jacobian_(SIG, ift, fun, J, None, c.traj(N)); c.note()             # without an x_next output
        assert_periodic(J, Jo, *tj, f"{name}/{dname} implicit IFT Jacobian, no y, N={N}")
    Zx, t2, dt2 = (np.concatenate([a, a[:1]]) for a in (c.Z, c.t, c.dt))
    Eo, J2o, J1o = c.om.implicit_midpoint_error(Zx, t2, dt2)
    got = RD.dynamics_error(fun, c.traj(MID)); c.note()
    assert_periodic(got, Eo, *((1e-4 * max(1.0, np.abs(Eo).max()), 1e-4) if c.f32 else (1e-10 * max(1.0, np.abs(Eo).max()), 0.0)), f"{name}/{dname} implicit dynamics_error")
    tdt = torch.float32 if c.f32 else torch.float64
    J2, J1 = RD.alloc_jacobians(fun, MID - 1, c.traj(BASE).data), RD.alloc_jacobians(fun, MID - 1, c.traj(BASE).data)
    y2 = torch.empty((MID - 1, c.n), dtype=tdt, device="cuda")
    RD.dynamics_error_jacobian_(SIG, RD.ForwardAD(), fun, J2, J1, y2, None, c.traj(MID)); c.note()
    tjj = (1e-4 * max(1.0, np.abs(J1o).max()), 1e-4) if c.f32 else (1e-10 * max(1.0, np.abs(J1o).max()), 0.0)
    assert_periodic(J1, J1o, *tjj, f"{name}/{dname} implicit dynamics_error_jacobian J1")
    assert_periodic(J2, J2o, *tjj, f"{name}/{dname} implicit dynamics_error_jacobian J2")
    # rollout (trajectory-major): latency-sized and chip-sized
    for M, K in ((70, 6), (8261, 3)):
        st, Zb, tb, dtb = _stack(c, M, K, False, seed=17)
        Zr = np.array(Zb)
        for k in range(K - 1):                                        # the oracle's Newton step, knot by knot (src/discrete_dynamics.jl:217-235)
            zk = Zr[:, k, :]
            xn, _, _ = c.om.implicit_midpoint(zk, tb[:, k], dtb[:, k], want_jacobian=False)
            Zr[:, k + 1, :c.n] = xn
        RD.rollout_(SIG, fun, st); c.note()
        s = max(1.0, float(np.abs(Zr).max()))
        assert_periodic(st["data"], Zr, *((5e-4 * s, 5e-4) if c.f32 else (1e-8 * s, 0.0)), f"{name}/{dname} implicit rollout M={M}")
        if M == 70:
            # the one-lane-per-trajectory form (RDHIP_TUNE rollout=plain; what stacks of 2^26 knots and more take): for the rigid bodies a different
            # pair of kernels from the cascade form above
            st2, _, _, _ = _stack(c, M, K, False, seed=17)
            RD.tune("rollout=plain")
            try:
                RD.rollout_(SIG, fun, st2); c.note()
            finally:
                RD.tune("reset")
            assert_periodic(st2["data"], Zr, *((5e-4 * s, 5e-4) if c.f32 else (1e-8 * s, 0.0)), f"{name}/{dname} implicit rollout, rollout=plain, M={M}")


# ------------------------------------------------------------------------------------------------------------------------------------------
# the mixed-model batch as ONE launch (csrc/mixed.hpp): satellite + cartpole under RK3 / RK4, fp32 / fp64, both placements of the cartpole's rollout
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tm", [True, False], ids=["time_major", "trajectory_major"])
@pytest.mark.parametrize("dname", ["f64", "f32"])
@pytest.mark.parametrize("rule", [RD.RK3, RD.RK4])
def test_sweep_mixed_model_call_in_one_launch(rule, dname, tm):
    ca, cb = case_of("sat_wd", dname), case_of("cartpole", dname)
    for idle in (0, 1):
        for MA, MB, K in ((132, 200, 9), (64, 64, 5)) + (((4096, 4096, 5),) if dname == "f32" and idle == 1 else ()):      # (the last: the two-per-CU form)
            RD.tune(f"mixed_idle={idle}")
            try:
                sa, Za, ta, da = _stack(ca, MA, K, tm, seed=23)
                sb, Zb, tb, db = _stack(cb, MB, K, tm, seed=29)
                funs = [RD.DiscretizedDynamics(ca.model, rule), RD.DiscretizedDynamics(cb.model, rule)]
                Js = [RD.alloc_jacobians(funs[0], MA * K, sa["data"]).fill_(float("nan")), RD.alloc_jacobians(funs[1], MB * K, sb["data"]).fill_(float("nan"))]
                call = RD.PreparedRolloutJacobianMulti(SIG, RD.ForwardAD(), funs, Js, [sa, sb], None, time_major=tm)
                call()
                assert call.launches == 1 and "rollout_jac_mixed_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
                ca.note()
            finally:
                RD.tune("reset")
            for c, st, Zbase, tbase, dbase, J, M in ((ca, sa, Za, ta, da, Js[0], MA), (cb, sb, Zb, tb, db, Js[1], MB)):
                nz = c.n + c.m
                Zr = c.om.rollout(ORULE[rule], Zbase, tbase, dbase)
                s = max(1.0, float(np.abs(Zr).max()))
                got = st["data"].transpose(0, 1) if tm else st["data"]
                assert_periodic(got, Zr, *((2e-4 * s, 2e-4) if c.f32 else (1e-10 * s, 0.0)), f"mixed {c.name}/{dname} states {rule.__name__} idle={idle} M={M} tm={tm}")
                Jo, _ = c.om.jacobian(ORULE[rule], O.FORWARD_AD, Zr.reshape(-1, nz), tbase.reshape(-1), dbase.reshape(-1))
                Jo = Jo.reshape(Zbase.shape[0], K, c.n, nz)
                sj = max(1.0, float(np.abs(Jo).max()))
                Jg = J.reshape(K, M, c.n, nz).transpose(0, 1) if tm else J.reshape(M, K, c.n, nz)
                assert_periodic(Jg, Jo, *((1e-3 * sj, 1e-3) if c.f32 else (1e-10 * sj, 0.0)), f"mixed {c.name}/{dname} Jacobians {rule.__name__} idle={idle} M={M} tm={tm}")


# ------------------------------------------------------------------------------------------------------------------------------------------
# the error-state kernels in fp32 (csrc/errstate.hip): every branch of rdhip_errstate_jacobian / rdhip_state_diff / rdhip_nabla_errstate_jacobian
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["sat_wd", "quad_mrp_b", "sat_rp_b", "cartpole"])
def test_sweep_error_state_kernels_fp32(name):
    c = case_of(name, "f32")
    n, e, N = c.n, RD.errstate_dim(c.model), 203
    X = c.Z[:, :n].astype(np.float32)
    Xd = torch.as_tensor(X).cuda()
    Go = c.om.errstate_jacobian(X.astype(np.float64))
    pattern = Go != 0
    for aligned in (True, False):                               # (an output that does not start on a 16-byte boundary takes the per-entry kernels)
        for clear in (True, False):
            buf = torch.full((N * n * e + 1,), 7.0, dtype=torch.float32, device="cuda")
            G = buf[(0 if aligned else 1):][:N * n * e].view(N, e, n).transpose(1, 2)
            RD.errstate_jacobian_(c.model, G, Xd, clear=clear); c.note()
            Gh = G.cpu().numpy().astype(np.float64)
            if clear or name == "cartpole":                     # (EuclideanState: always the whole identity, src/statevectortype.jl:149-155)
                np.testing.assert_allclose(Gh, Go, rtol=0, atol=2e-6, err_msg=f"{name} errstate_jacobian clear aligned={aligned}")
            else:
                np.testing.assert_allclose(Gh[pattern], Go[pattern], rtol=0, atol=2e-6, err_msg=f"{name} errstate_jacobian pattern aligned={aligned}")
                if name == "sat_wd":                            # the quaternion pattern kernel touches nothing else
                    assert np.all(Gh[~pattern] == 7.0)
    if name == "cartpole":
        return
    X0 = np.roll(X, 1, axis=0)
    # state_diff: contiguous states (the tile kernel) and states that are columns of a wide array (leading dimension 40: the per-entry kernel)
    do = c.om.state_diff(X.astype(np.float64), X0.astype(np.float64))
    ok = np.abs(do).max(axis=1) < 30.0                          # (Cayley v / w near w = 0: ill-conditioned in fp32, tests/test_gpu_round6.py)
    got = RD.state_diff(c.model, Xd, torch.as_tensor(X0).cuda()); c.note()
    np.testing.assert_allclose(got.cpu().numpy().astype(np.float64)[ok], do[ok], rtol=2e-4, atol=2e-5)
    wide, wide0 = torch.zeros((N, 40), dtype=torch.float32, device="cuda"), torch.zeros((N, 40), dtype=torch.float32, device="cuda")
    wide[:, :n] = Xd; wide0[:, :n] = torch.as_tensor(X0).cuda()
    got = RD.state_diff(c.model, wide[:, :n], wide0[:, :n]); c.note()
    np.testing.assert_allclose(got.cpu().numpy().astype(np.float64)[ok], do[ok], rtol=2e-4, atol=2e-5)
    # nabla errstate Jacobian
    Xb = np.roll(X, 2, axis=0)
    Ho = c.om.nabla_errstate_jacobian(X.astype(np.float64), Xb.astype(np.float64))
    for clear in (True, False):
        dG = torch.zeros((N, e, e), dtype=torch.float32, device="cuda").transpose(1, 2)
        RD.nabla_errstate_jacobian_(c.model, dG, Xd, torch.as_tensor(Xb).cuda(), clear=clear); c.note()
        np.testing.assert_allclose(dG.cpu().numpy().astype(np.float64), Ho, rtol=2e-5, atol=2e-5, err_msg=f"{name} nabla clear={clear}")
