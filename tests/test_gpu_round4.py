"""GPU tests added in round 4: the whole of BASELINE config 5 on ONE GPU, ordering of the overlap mode against the multi-model call and graph
launches, the fused rollout -> Jacobians launch on shapes whose tiles share 128-byte lines, dispatch checked by kernel IDENTITY
(rdhip_last_kernel_name), and the multi-process control path of bench.py on a 1-GPU box."""
import json
import os
import subprocess
import sys

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


def _tm_stack(data, t, dt, cast):
    """(M, K, nz) host arrays -> the time-major device stack {"data" (K, M, nz), "t", "dt" (K, M)}"""
    return {"data": torch.as_tensor(cast(np.ascontiguousarray(data.transpose(1, 0, 2)))).cuda(),
            "t": torch.as_tensor(np.ascontiguousarray(t.T)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T)).cuda()}


def test_config5_whole_workload_on_one_gpu_in_one_call():
    """BASELINE configs[4] as ONE GPU sees the whole of it: 8 192 trajectories x 256 knots, even trajectories Cartpole / odd Satellite (4 096 each),
    fp32, rollouts and the Jacobians along them in ONE rdhip_rollout_jacobian_multi call (rollout!: src/trajectories.jl:436-441, jacobian!:
    src/functionbase.jl:242-257).  At this size the checks are: sampled trajectories and their Jacobians against the fp64 oracle (rel 2e-3: fp32
    error accumulated over 255 steps), the linearisation identity x_{k+1} = f(z_k) over the WHOLE batch (discrete dynamics of every simulated knot
    point = the next simulated state, to fp32 rounding), and the Jacobians of the one call against a separate rdhip_jacobian call on the simulated
    states over the whole batch (the same column code: last-bit agreement)."""
    rng = np.random.default_rng(55)
    M, K = 4096, 256
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    f32 = lambda a: a.astype(np.float32)     # noqa: E731
    specs = [(RD.Cartpole(), O.cartpole(), 1, lambda n: 0.2 * rng.random((n, 5))),
             (RD.Satellite(), O.satellite(), 6, lambda n: 0.3 * H.rand_rigid(rng, n, 6) + np.array([0] * 3 + [0.7, 0, 0, 0] + [0] * 12))]
    funs, stacks, Js, keep = [], [], [], []
    t = np.tile(np.arange(K) * 0.01, (M, 1)); dt = np.full((M, K), 0.01)
    for model, om, mctl, gen in specs:
        n = RD.state_dim(model)
        data = gen(M * K).reshape(M, K, n + mctl)
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        st = _tm_stack(data, t, dt, f32)
        funs.append(dm); stacks.append(st); Js.append(RD.alloc_jacobians(dm, K * M, st["data"]).fill_(7.0)); keep.append((model, om, mctl, n, f32(data)))
    RD.rollout_jacobian_multi_(sig, AD, funs, Js, stacks, None, time_major=True)
    torch.cuda.synchronize()
    sample = [0, 1, 63, 64, 2047, 4095]
    for p, (model, om, mctl, n, data) in enumerate(keep):
        nz = n + mctl
        rolled = host(stacks[p]["data"])                      # (K, M, nz)
        assert np.isfinite(rolled).all()
        np.testing.assert_array_equal(rolled[..., n:], data.transpose(1, 0, 2)[..., n:])        # controls untouched
        np.testing.assert_array_equal(rolled[0, :, :n], data[:, 0, :n])                            # initial states kept
        ref = om.rollout(O.RK4, data[sample].astype(np.float64), t[sample], dt[sample])
        np.testing.assert_allclose(rolled[:, sample].transpose(1, 0, 2)[..., :n], ref[..., :n], rtol=2e-3, atol=2e-3)
        # x_{k+1} = f(z_k) over the whole batch
        Zflat = RD.PackedTrajectory(n, mctl, stacks[p]["data"].reshape(K * M, nz), stacks[p]["t"].reshape(-1), stacks[p]["dt"].reshape(-1))
        Y = torch.empty((K * M, n), dtype=torch.float32, device="cuda")
        RD.evaluate_(funs[p], Y, Zflat)
        Yh = host(Y).reshape(K, M, n)
        np.testing.assert_allclose(Yh[:-1], rolled[1:, :, :n], rtol=1e-5, atol=2e-6)
        # the Jacobians of the one call = those of a separate call on the simulated states, over the whole batch
        J2 = RD.alloc_jacobians(funs[p], K * M, stacks[p]["data"])
        RD.jacobian_(sig, AD, funs[p], J2, None, Zflat)
        d = (Js[p] - J2).abs().max().item()
        scale = J2.abs().max().item()
        assert d <= 2e-6 * max(1.0, scale), (type(model).__name__, d, scale)
        del J2
        # ... and the oracle's at sampled trajectories
        Jg = host(Js[p]).reshape(K, M, n, nz)
        for j in (0, 64, 4095):
            Jo, _ = om.jacobian(O.RK4, O.FORWARD_AD, rolled[:, j].astype(np.float64), t[j], dt[j])
            np.testing.assert_allclose(Jg[:, j], Jo, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("kind", ["satellite", "satfull", "quadrotor"])
@pytest.mark.parametrize("rtag", ["mrp", "rp"])
@pytest.mark.parametrize("ftag", ["world", "body"])
@pytest.mark.parametrize("where", ["host", "cuda"])
def test_error_state_expansion_of_three_parameter_rigid_bodies(kind, rtag, ftag, where):
    """rdhip_error_expansion for RigidBody{MRP} / RigidBody{RodriguesParam}: E_k = G(x_{k+1})' [A_k G(x_k) | B_k] with the 3 x 3 blocks of
    src/liestate.jl:262-298 (the reference generates errstate_jacobian! for any R) against exp_*_{mrp,rp}_*.npz -- G(x_{k+1})' d f_d(x_k (+) delta, u) /
    d(delta, u) differentiated from the composition map itself (tests/golden/make_golden_rot3.py) -- and against the oracle's product; dynamics_error too."""
    name = f"{kind}_{rtag}_{ftag}"
    d, g = H.load("dyn_" + name), H.load("exp_" + name)
    R = RD.MRP if rtag == "mrp" else RD.RodriguesParam
    if kind == "quadrotor":
        model = RD.Quadrotor(R, mass=float(d["mass"]), J=d["J"], gravity=d["gravity"], motor_dist=float(d["L"]), kf=float(d["kf"]), km=float(d["km"]), bodyframe=bool(d["frame"]))
    else:
        model = RD.Satellite(float(d["mass"]), d["J"], R, bodyframe=bool(d["frame"]))
    om = H.oracle_model(name, d)
    n, m, e = RD.state_dim(model), RD.control_dim(model), RD.errstate_dim(model)
    assert (n, e) == (12, 12)
    N = d["Z"].shape[0]
    for Q, rn, orule in ((RD.RK4, "rk4", O.RK4), (RD.RK3, "rk3", O.RK3)):
        dm = RD.DiscretizedDynamics(model, Q)
        Z = RD.PackedTrajectory(n, m, d["Z"], d["t"], d["dt"])
        if where == "cuda":
            Z = Z.to("cuda")
        E = RD.alloc_error_expansions(dm, N, Z.data)
        y = RD.evaluate(dm, Z)[: N - 1] * 0
        RD.error_expansion_(dm, E, y, Z)
        scale = max(1.0, np.abs(g["E_" + rn]).max())
        np.testing.assert_allclose(host(E), g["E_" + rn], rtol=0, atol=TOL * scale)
        Eo, yo = om.error_expansion(orule, d["Z"], d["t"], d["dt"])
        np.testing.assert_allclose(host(E), Eo, rtol=0, atol=TOL * scale)
        np.testing.assert_allclose(host(y), yo, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(RD.dynamics_error(dm, Z)), g["err_" + rn], rtol=0, atol=TOL)


def test_overlap_mode_orders_the_multi_model_call_and_graph_launches_behind_the_internal_stream():
    """ORDERING RULE of rdhip_ctx_set_overlap (include/rdhip.h): every other operation on the context is ordered behind everything issued before it.
    An rdhip_jacobian that went to the INTERNAL stream is followed (a) by rdhip_rollout_jacobian_multi whose child streams write the same J, and
    (b) by rdhip_graph_launch of a captured call that writes the same J: in both cases the later operation's values must be the final ones."""
    from rdhip import _lib as L
    rng = np.random.default_rng(91)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    sat, cart = RD.Satellite(), RD.Cartpole()
    dsat, dcart = RD.DiscretizedDynamics(sat, RD.RK4), RD.DiscretizedDynamics(cart, RD.RK4)
    M, K = 192, 120
    N = M * K
    ctx = L.Context(0)
    ctx.set_overlap(True)
    t = np.tile(np.arange(K) * 0.02, (M, 1)); dt = np.full((M, K), 0.02)
    # the multi call: cartpole = model 0 (the context's own stream), satellite = model 1 (a child stream)
    dsatd = H.rand_rigid(rng, N, 6).reshape(M, K, 19)
    dcartd = rng.random((N, 5)).reshape(M, K, 5)
    ident = lambda a: a     # noqa: E731
    st_sat, st_cart = _tm_stack(dsatd, t, dt, ident), _tm_stack(dcartd, t, dt, ident)
    Jsat, Jcart = RD.alloc_jacobians(dsat, N, st_sat["data"]), RD.alloc_jacobians(dcart, N, st_cart["data"])
    multi = RD.PreparedRolloutJacobianMulti(sig, AD, [dcart, dsat], [Jcart, Jsat], [st_cart, st_sat], None, time_major=True, ctx=ctx)
    # reference result of the multi call (run alone)
    multi(); ctx.synchronize()
    Jsat_ref, Jcart_ref = Jsat.clone(), Jcart.clone()
    st_ref = st_sat["data"].clone()
    # two plain Jacobian calls on OTHER inputs that write the multi call's satellite J: the second of a pair goes to the internal stream
    other = [RD.PackedTrajectory(13, 6, H.rand_rigid(rng, N, 6), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda") for _ in range(2)]
    Jdummy = RD.alloc_jacobians(dsat, N, st_sat["data"])
    first = RD.prepare_jacobian(sig, AD, dsat, Jdummy, None, other[0], ctx=ctx)
    second = RD.prepare_jacobian(sig, AD, dsat, Jsat, None, other[1], ctx=ctx)       # writes the multi call's output
    for rep in range(6):
        st_sat["data"].copy_(torch.as_tensor(np.ascontiguousarray(dsatd.transpose(1, 0, 2))).cuda()); st_cart["data"].copy_(torch.as_tensor(np.ascontiguousarray(dcartd.transpose(1, 0, 2))).cuda())
        torch.cuda.synchronize()
        first(); second()                                 # `second` is in flight on the internal stream ...
        multi()                                           # ... when the multi call forks its child streams
        ctx.synchronize()
        assert torch.equal(Jsat, Jsat_ref), f"rep {rep}: a Jacobian call on the internal stream overwrote the multi-model call's output"
        assert torch.equal(Jcart, Jcart_ref) and torch.equal(st_sat["data"], st_ref)
    # (b) graph launch: a captured Jacobian call writing Jsat, replayed right behind an internal-stream call that writes Jsat as well
    ctx.set_overlap(False)
    third = RD.prepare_jacobian(sig, AD, dsat, Jsat, None, other[0], ctx=ctx)
    third(); ctx.synchronize()
    J_third = Jsat.clone()
    ctx.begin_capture(); third(); g = ctx.end_capture()
    ctx.set_overlap(True)
    for rep in range(6):
        first(); second()
        ctx.launch(g)
        ctx.synchronize()
        assert torch.equal(Jsat, J_third), f"rep {rep}: the graph launch was not ordered behind the internal stream"
    g.close()
    ctx.set_overlap(False)
    ctx.close()


@pytest.mark.parametrize("shape", [(66, 101, True), (66, 101, False), (130, 37, True), (4, 101, False)])
def test_fused_rollout_jacobians_when_tiles_share_cache_lines(shape):
    """The fused rollout -> Jacobians launch (csrc/fused.hpp) on stacks whose tiles do NOT start on 128-byte lines (M not a multiple of 64 time-major;
    K (n+m) 8 not a multiple of 128 trajectory-major), with few Jacobian workgroups so that one compute unit reads neighbouring tiles one after the
    other: a consumer must not serve the later tile from a line it cached before the producer wrote that tile's part (the poll is followed by an
    agent-scope acquire).  Against the oracle and against the two-call path, repeated."""
    M, K, tm = shape
    rng = np.random.default_rng(17)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    model, om = RD.Satellite(), O.satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    data = H.rand_rigid(rng, M * K, 6).reshape(M, K, 19)
    t = np.tile(np.arange(K) * 0.02, (M, 1)); dt = np.full((M, K), 0.02)
    ref = om.rollout(O.RK4, data, t, dt)
    Jref = om.jacobian(O.RK4, O.FORWARD_AD, ref.reshape(M * K, -1), t.reshape(-1), dt.reshape(-1))[0].reshape(M, K, 13, 19)
    mk = (lambda: _tm_stack(data, t, dt, lambda a: a)) if tm else (lambda: {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(t).cuda(), "dt": torch.as_tensor(dt).cuda()})
    try:
        for nj in ("3", "16", "0"):                                    # rdhip_tune: the dispatcher's knobs at run time (include/rdhip.h); 0 = every CU the rollout leaves
            RD.tune("fused_jac_workgroups=" + nj)
            for rep in range(3):
                st = mk()
                J = RD.alloc_jacobians(dm, M * K, st["data"]).fill_(-5.0)
                launches = RD.rollout_jacobian_(sig, AD, dm, J, st, None, time_major=tm)
                if tm and (M * 13 * 19 * 8) % 16 == 0 or not tm:
                    assert launches == 1, (shape, nj, launches)
                got = host(st["data"]); Jg = host(J).reshape((K, M, 13, 19) if tm else (M, K, 13, 19))
                if tm:
                    got, Jg = got.transpose(1, 0, 2), Jg.transpose(1, 0, 2, 3)
                np.testing.assert_allclose(got, ref, rtol=0, atol=TOL)
                np.testing.assert_allclose(Jg, Jref, rtol=0, atol=TOL * max(1.0, np.abs(Jref).max()))
    finally:
        RD.tune("reset")


def test_two_fused_launches_from_two_contexts_do_not_hang():
    """Two contexts, each issuing fused rollout -> Jacobians launches on its own stream at the same time: the library orders every fused launch behind
    the previous one on the device (csrc/api.hip fused_launch), because two spin-waiting launches in flight could starve each other's producers.
    Run in a child process with a time limit; results against the oracle."""
    code = r'''
import sys, numpy as np, torch
sys.path[:0] = [%r, %r]
import rdhip as RD
from rdhip import _lib as L
from oracle import rd_oracle as O
from tests import helpers as H
rng = np.random.default_rng(3)
M, K = 4096, 40
dm = RD.DiscretizedDynamics(RD.Satellite(), RD.RK4)
om = O.satellite()
streams = [torch.cuda.Stream() for _ in range(2)]
work = []
for s in streams:
    data = H.rand_rigid(rng, M * K, 6).reshape(M, K, 19); t = np.tile(np.arange(K) * 0.02, (M, 1)); dt = np.full((M, K), 0.02)
    st = {"data": torch.as_tensor(np.ascontiguousarray(data.transpose(1, 0, 2))).cuda(), "t": torch.as_tensor(np.ascontiguousarray(t.T)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T)).cuda()}
    work.append((s, data, t, dt, st, RD.alloc_jacobians(dm, M * K, st["data"])))
torch.cuda.synchronize()
for rep in range(10):
    for s, data, t, dt, st, J in work:
        with torch.cuda.stream(s):
            st["data"].copy_(torch.as_tensor(np.ascontiguousarray(data.transpose(1, 0, 2))).cuda(), non_blocking=True)
            assert RD.rollout_jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, st, None, time_major=True) == 1
torch.cuda.synchronize()
for s, data, t, dt, st, J in work:
    ref = om.rollout(O.RK4, data[:3], t[:3], dt[:3])
    np.testing.assert_allclose(st["data"].cpu().numpy().transpose(1, 0, 2)[:3], ref, rtol=0, atol=1e-10)
print("OK")
''' % (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_dispatch_reaches_the_intended_kernels():
    """A guard against silent dispatch regressions (a template condition once sent the cartpole to the generic unrolled tile kernel: 176 instead of 41 us
    at 1 M knot points, every parity test still green) -- by IDENTITY: rdhip_last_kernel_name says which kernel a call was dispatched to."""
    rng = np.random.default_rng(5)
    sig, AD, FD, UD = RD.StaticReturn(), RD.ForwardAD(), RD.FiniteDifference(), RD.UserDefined()
    ift = RD.ImplicitFunctionTheorem(RD.ForwardAD())
    sat, cart, quad = RD.Satellite(), RD.Cartpole(), RD.Quadrotor()
    rigid = lambda N, m=6: H.rand_rigid(rng, N, m)     # noqa: E731
    cases = [(cart, RD.RK4, AD, rng.random((1 << 16, 5)), np.float64, ["jac_ad_small_kernel"]),
             (cart, RD.RK4, AD, rng.random((1 << 16, 5)), np.float32, ["jac_ad_small_kernel", "float"]),
             (sat, RD.RK4, AD, rigid(1 << 16), np.float64, ["jac_ad_split_kernel", "TablePlanS<19, 4"]),
             (sat, RD.RK4, AD, rigid(1 << 17), np.float32, ["jac_ad_split_kernel", "float"]),
             (sat, RD.RK4, AD, rigid(1000), np.float64, ["jac_ad_colsplit_kernel"]),
             (RD.Satellite(bodyframe=True), RD.RK4, AD, rigid(1 << 16), np.float64, ["jac_ad_split_kernel"]),
             (quad, RD.RK4, AD, rigid(1 << 16, 4), np.float64, ["jac_ad_split_kernel", "TablePlanS<17, 4"]),
             (RD.Quadrotor(bodyframe=True), RD.RK4, AD, rigid(1 << 16, 4), np.float64, ["jac_ad_split_kernel"]),
             (sat, RD.RK4, FD, rigid(1 << 16), np.float64, ["jac_fd_cascade_kernel"]),
             (sat, RD.RK4, UD, rigid(1 << 16), np.float64, ["jac_chain_split_kernel"]),
             (sat, RD.ImplicitMidpoint, ift, rigid(1 << 16), np.float64, ["implicit_ift_split_kernel"]),
             (RD.LieSatellite(), RD.RK4, AD, rng.random((1 << 16, 10)) + 0.1, np.float64, ["jac_ad_split_kernel"]),
             (RD.SatellitePair(), RD.RK4, AD, rng.random((1 << 16, 20)) + 0.1, np.float64, ["jac_ad_split_kernel"])]
    for model, rule, diff, Zn, dtype, needles in cases:
        N = Zn.shape[0]
        n, m = RD.state_dim(model), RD.control_dim(model)
        dm = RD.DiscretizedDynamics(model, rule)
        Z = RD.PackedTrajectory(n, m, np.ascontiguousarray(Zn, dtype=dtype), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda")
        J = RD.alloc_jacobians(dm, N, Z.data)
        Y = torch.empty((N, n), dtype=Z.data.dtype, device="cuda")
        RD.jacobian_(sig, diff, dm, J, Y, Z)
        name = RD.last_kernel_name()
        for needle in needles:
            assert needle in name, (type(model).__name__, rule.__name__, type(diff).__name__, dtype.__name__, N, "dispatched to", name)
    torch.cuda.synchronize()
    # the other families
    X = torch.as_tensor(rigid(1 << 18, 4)[:, :13]).cuda()
    G = RD.alloc_errstate_jacobians(quad, 1 << 18, X)
    RD.errstate_jacobian_(quad, G, X, clear=True)
    assert "errstate_tile_kernel" in RD.last_kernel_name(), RD.last_kernel_name()
    torch.cuda.synchronize()


def _run_bench(args, nproc, timeout=420):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", "29613",
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_multi_process_control_path_on_one_gpu(backend):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), both ranks on cuda:0 (--same-device): the N > 1
    control path -- rendezvous, barriers, max-over-ranks timing, rank 0's ONE JSON line with n_gpus = 2 and the whole-job value -- must work before a
    real node ever runs it.  gloo: the CPU control path.  nccl: two ranks on ONE device cannot form an RCCL communicator; bench.py must fall back to
    the gloo control path (or report an error field) and still print its line instead of hanging."""
    r, lines = _run_bench(["--same-device", "--backend", backend, "--steps", "5", "--warmup", "2", "--gather", "none", "--no-extras", "--no-cpu-baseline",
                           "--prewarm-ms", "20", "--phase-timeout", "120"], 2)
    assert len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads(lines[0])
    if backend == "nccl" and "error" in out:
        return                                             # the watchdog's line: an error field instead of a hang is the accepted outcome here
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak"
    if backend == "gloo":
        assert "error" not in out, out.get("error")
        assert out["value"] > 0 and out["config"]["knot_points_total"] == 2 * out["config"]["knot_points_per_gpu"]
        assert "sharded_configs" not in out and "gather" not in out
        assert abs(out["value"] - 2 * 65536 * 5 / (out["ms_per_step"] * 1e-3 * 5)) < 1e-6 * out["value"]
    else:
        assert out.get("value") or out.get("error")


def test_device_memory_outlives_its_context():
    """rdhip_free(NULL, ptr): a garbage-collected host (the Julia module's finalizers run in no particular order) releases device memory after the
    context it was allocated through has been destroyed."""
    import ctypes as C
    from rdhip import _lib as L
    lib = L.lib()
    h = C.c_void_p()
    assert lib.rdhip_ctx_create(0, None, L.STREAM_OWN, C.byref(h)) == 0
    p = C.c_void_p()
    assert lib.rdhip_malloc(h, C.byref(p), 1 << 20) == 0 and p.value
    assert lib.rdhip_ctx_destroy(h) == 0
    assert lib.rdhip_free(None, p) == 0


@pytest.mark.parametrize("where", ["host", "cuda"])
def test_wrenches_of_a_custom_rigid_body(where):
    """wrenches(model, x, u, t) and wrench_jacobian!(F, model, z) work for ANY RigidBody that defines forces and moments (src/rigidbody.jl:244-248,
    :315-320): for RD.CustomRigidBody -- the two functions given as HIP source -- the host composes them from the same snippets (a helper model whose
    first six outputs are [F; M]).  The quadrotor of test/quadrotor.jl:56-96 written that way against the oracle's forward duals and against the
    compiled-in quadrotor's rdhip_wrenches / rdhip_wrench_jacobian."""
    from examples.custom_models import QUADROTOR_FORCES_SRC as forces, QUADROTOR_MOMENTS_SRC as moments
    rng = np.random.default_rng(74)
    model = RD.CustomRigidBody(4, forces, moments, mass=0.5, J=np.diag([0.0023, 0.0023, 0.004]), params=(0.0, 0.0, -9.81, 0.1750, 1.0, 0.0245))
    ref_model, om = RD.Quadrotor(), O.quadrotor()
    for N in (130, 1):
        Zn = H.rand_rigid(rng, N, 4, 0.5, 1.5)
        Zn[0, 13] = -0.3                                             # max(0, kf u): a rotor command below zero
        t, dt = np.arange(N) * 0.1, np.full(N, 0.1); dt[-1] = 0.0   # (a terminal knot point: wrenches take the control as given)
        Z = RD.PackedTrajectory(13, 4, Zn, t, dt)
        if where == "cuda":
            Z = Z.to("cuda")
        W, F = RD.wrenches(model, Z), RD.wrench_jacobian(model, Z)
        Wo, Fo = om.wrenches(Zn, t), om.wrench_jacobian(Zn, t)
        assert tuple(W.shape) == (N, 6) and tuple(F.shape) == (N, 6, 17)
        np.testing.assert_allclose(host(W), Wo, rtol=0, atol=TOL)
        np.testing.assert_allclose(host(F), Fo, rtol=0, atol=TOL * max(1.0, np.abs(Fo).max()))
        np.testing.assert_allclose(host(W), host(RD.wrenches(ref_model, Z)), rtol=0, atol=TOL)
        np.testing.assert_allclose(host(F), host(RD.wrench_jacobian(ref_model, Z)), rtol=0, atol=TOL)
    with pytest.raises(NotImplementedError):
        RD.wrenches(RD.CustomModel(4, 1, H.CARTPOLE_SRC, params=(1.0, 0.2, 0.5, 9.81)), RD.PackedTrajectory(4, 1, np.zeros((2, 5)), 0.0, 0.1))


def test_batch_table_slots_are_reused_safely_without_an_event_per_launch():
    """rdhip_jacobian_multi with more than 32 trajectories keeps its batch table on the device, in one of 8 slots per context; a call that finds its
    table there (a solver calling with the same arrays again) records no event behind its launch, so a slot may be due for overwriting while launches
    that read it are still queued -- the library must let them finish first.  20 different tables (three slot generations), each called three times
    back to back with no synchronisation in between, every result against one rdhip_jacobian call per trajectory."""
    rng = np.random.default_rng(404)
    model = RD.Satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    nb, per, ntab = 40, 101, 20
    n = RD.state_dim(model)
    tables = []
    for k in range(ntab):
        Zs = [RD.PackedTrajectory(n, 6, H.rand_rigid(rng, per, 6), 0.1 * b + np.arange(per) * 0.05, np.full(per, 0.05)).to("cuda") for b in range(nb)]
        Js = [RD.alloc_jacobians(dm, per, Zs[0].data).fill_(7.0) for _ in range(nb)]
        Ys = [torch.full((per, n), 7.0, dtype=torch.float64, device="cuda") for _ in range(nb)]
        tables.append((Zs, Js, Ys))
    for Zs, Js, Ys in tables:                       # no synchronisation anywhere in this loop
        for _ in range(3):
            assert RD.jacobian_multi_(sig, AD, dm, Js, Ys, Zs) == 1
    torch.cuda.synchronize()
    for Zs, Js, Ys in tables:
        for b in (0, 17, nb - 1):
            Jr = RD.alloc_jacobians(dm, per, Zs[b].data)
            Yr = torch.empty((per, n), dtype=torch.float64, device="cuda")
            RD.jacobian_(sig, AD, dm, Jr, Yr, Zs[b])
            # (101 knots alone take the column-split kernel: same columns, another order of operations -- not bit-equal)
            np.testing.assert_allclose(host(Js[b]), host(Jr), rtol=0, atol=1e-13)
            np.testing.assert_allclose(host(Ys[b]), host(Yr), rtol=0, atol=1e-13)


def test_dynamic_tile_schedule_gives_what_the_static_one_gives():
    """From 8 tiles per workgroup on (131 072 knots on a whole MI355X) the split-role Jacobian kernels take a workgroup's fourth and later tiles from a
    device-wide counter (kernels.hpp jac_split_loop; two counter words per launch from a ring of the context, reset by the launch's last workgroup).
    Knot points are independent, so a big batch must give bit for bit what its two halves give in separate calls (64-knot tiles fall on the same
    knots; the halves are below the threshold: static schedule) -- for every kernel that goes through that loop, with a ragged last tile, in
    overlap mode (two streams, a ring each), and after more launches than the ring has slots (a counter that was not reset would skip tiles)."""
    rng = np.random.default_rng(808)
    sig, AD, FD, UD = RD.StaticReturn(), RD.ForwardAD(), RD.FiniteDifference(), RD.UserDefined()
    half = 8 * 256 * 64 // 2
    N = 2 * half + 37
    custom = RD.CustomModel(13, 6, H.SATELLITE_SRC, params=(2.0, 2.0, 3.0, 1.0), liestate=(RD.QuatRotation, (3, 6)))
    cases = [(RD.Satellite(), 6, AD, np.float64), (RD.Satellite(), 6, FD, np.float64), (RD.Satellite(), 6, UD, np.float64), (RD.Quadrotor(), 4, AD, np.float64),
             (RD.Satellite(bodyframe=True), 6, AD, np.float32), (custom, 6, AD, np.float64)]
    for model, m, diff, dtype in cases:
        n = RD.state_dim(model)
        dm = RD.DiscretizedDynamics(model, RD.RK4)
        Zn = np.ascontiguousarray(H.rand_rigid(rng, N, m), dtype=dtype)
        t, dt = np.arange(N) * 0.05, np.full(N, 0.05)
        whole = RD.PackedTrajectory(n, m, Zn, t, dt).to("cuda")
        J = [RD.alloc_jacobians(dm, N, whole.data) for _ in range(2)]
        Y = [torch.empty((N, n), dtype=whole.data.dtype, device="cuda") for _ in range(2)]
        for i in range(70):                                   # (more than twice the 32 slots of the context's ring)
            RD.jacobian_(sig, diff, dm, J[i & 1], Y[i & 1], whole)
        name = RD.last_kernel_name().split("(")[0].rstrip()
        if dtype is np.float64 and model is not custom:       # (fp32 and hipRTC models keep the static schedule)
            assert "[dynamic tile schedule]" in RD.last_kernel_name(), name     # the launch took the dynamic schedule (rdhip_last_kernel_name's marker)
        parts = []
        for lo, hi in ((0, half), (half, N)):
            Zp = RD.PackedTrajectory(n, m, Zn[lo:hi], t[lo:hi], dt[lo:hi]).to("cuda")
            Jp = RD.alloc_jacobians(dm, hi - lo, Zp.data)
            Yp = torch.empty((hi - lo, n), dtype=Zp.data.dtype, device="cuda")
            RD.jacobian_(sig, diff, dm, Jp, Yp, Zp)
            assert "[dynamic tile schedule]" not in RD.last_kernel_name()      # (4 tiles per workgroup: static)
            parts.append((Jp, Yp))
        torch.cuda.synchronize()
        for k in range(2):
            assert torch.equal(J[k], torch.cat([parts[0][0], parts[1][0]])), (type(model).__name__, type(diff).__name__, dtype.__name__)
            assert torch.equal(Y[k], torch.cat([parts[0][1], parts[1][1]]))
    # overlap mode: launches alternate between two streams, each with its own ring of counter words
    model, dm = RD.Satellite(), RD.DiscretizedDynamics(RD.Satellite(), RD.RK4)
    Zn = H.rand_rigid(rng, N, 6)
    whole = RD.PackedTrajectory(13, 6, Zn, np.arange(N) * 0.05, np.full(N, 0.05)).to("cuda")
    from rdhip import _lib as L
    ctx = L.Context(0)
    ctx.set_overlap(True)
    if True:
        Js = [RD.alloc_jacobians(dm, N, whole.data) for _ in range(4)]
        calls = [RD.prepare_jacobian(sig, AD, dm, Js[b], None, whole, ctx=ctx) for b in range(4)]
        for i in range(80):
            calls[i & 3]()
        ctx.synchronize()
        ref = RD.alloc_jacobians(dm, N, whole.data)
        RD.jacobian_(sig, AD, dm, ref, None, whole)
        torch.cuda.synchronize()
        for b in range(4):
            assert torch.equal(Js[b], ref)


def test_dynamic_tile_schedule_with_a_batch_table():
    """rdhip_jacobian_multi on enough tiles for the dynamic schedule (>= 8 per workgroup, trajectories of >= 16 tiles on average): 70 equal
    trajectories (one look-up per tile) and 72 ragged ones (two), the table on the device.  Every launch of a run of 40 (the context's ring of counter
    words wraps) must leave the same bits, and sampled trajectories must agree with one rdhip_jacobian call each."""
    rng = np.random.default_rng(909)
    model = RD.Satellite()
    dm = RD.DiscretizedDynamics(model, RD.RK4)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    n = RD.state_dim(model)
    for sizes in ([4096] * 70, [4096, 2000, 3001] * 24):
        Zs = [RD.PackedTrajectory(n, 6, H.rand_rigid(rng, N, 6), 0.1 * (b % 7) + np.arange(N) * 0.05, np.full(N, 0.05)).to("cuda") for b, N in enumerate(sizes)]
        Js = [[RD.alloc_jacobians(dm, N, Zs[0].data) for N in sizes] for _ in range(2)]
        Ys = [[torch.empty((N, n), dtype=torch.float64, device="cuda") for N in sizes] for _ in range(2)]
        for i in range(40):
            assert RD.jacobian_multi_(sig, AD, dm, Js[i & 1], Ys[i & 1], Zs) == 1
        name = RD.last_kernel_name().split("(")[0].rstrip()
        assert "jac_split_multi_kernel" in name and "[dynamic tile schedule]" in RD.last_kernel_name(), name      # the launch took the dynamic schedule
        torch.cuda.synchronize()
        for b in range(len(sizes)):
            assert torch.equal(Js[0][b], Js[1][b]) and torch.equal(Ys[0][b], Ys[1][b]), b
        for b in list(range(0, len(sizes), 9)) + [len(sizes) - 1]:
            Jr = RD.alloc_jacobians(dm, sizes[b], Zs[b].data)
            Yr = torch.empty((sizes[b], n), dtype=torch.float64, device="cuda")
            RD.jacobian_(sig, AD, dm, Jr, Yr, Zs[b])
            np.testing.assert_allclose(host(Js[0][b]), host(Jr), rtol=0, atol=1e-13)
            np.testing.assert_allclose(host(Ys[0][b]), host(Yr), rtol=0, atol=1e-13)


def test_tile_counters_per_xcd_range_with_stealing():
    """From 128 tiles per workgroup on (2 M knot points on a whole chip) every XCD walks its own eighth of the tiles with a counter of its own, and a
    workgroup whose range is used up takes tiles from the next XCD's range (kernels.hpp jac_split_loop, XR).  2 097 189 satellite knots (a ragged
    last tile) must give bit for bit what the two halves give in separate calls (64 tiles per workgroup: one counter over the plain order), three
    launches in a row (all ten counter words are reset by the last workgroup of a launch)."""
    rng = np.random.default_rng(1212)
    model, dm = RD.Satellite(), RD.DiscretizedDynamics(RD.Satellite(), RD.RK4)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    half = 128 * 256 * 64 // 2
    N = 2 * half + 37
    Zn = H.rand_rigid(rng, N, 6)
    t, dt = np.arange(N) * 0.05, np.full(N, 0.05)
    whole = RD.PackedTrajectory(13, 6, Zn, t, dt).to("cuda")
    J = RD.alloc_jacobians(dm, N, whole.data)
    Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    for _ in range(3):
        RD.jacobian_(sig, AD, dm, J, Y, whole)
    name = RD.last_kernel_name()
    assert "jac_ad_split_kernel" in name and name.split("(")[0].rstrip().endswith("true>") and "[dynamic tile schedule]" in name, name      # the instantiation WITH the dynamic schedule
    for lo, hi in ((0, half), (half, N)):
        Zp = RD.PackedTrajectory(13, 6, Zn[lo:hi], t[lo:hi], dt[lo:hi]).to("cuda")
        Jp = RD.alloc_jacobians(dm, hi - lo, Zp.data)
        Yp = torch.empty((hi - lo, 13), dtype=torch.float64, device="cuda")
        RD.jacobian_(sig, AD, dm, Jp, Yp, Zp)
        torch.cuda.synchronize()
        assert torch.equal(J[lo:hi], Jp) and torch.equal(Y[lo:hi], Yp)
        del Jp, Yp, Zp
