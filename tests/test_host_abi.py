"""CPU: the C-ABI library loads and exports every symbol include/rdhip.h declares; integer queries are bit-exact
against the oracle and the reference's index tables; host-side knot-point / trajectory logic; no CPU fallback."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import rdhip as RD
from oracle import rd_oracle as O
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(RD.HEADER_PATH).read()
    declared = sorted(set(re.findall(r"\b(rdhip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(RD.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in rdhip.h but not exported"
    assert sorted(RD.EXPORTS) == declared, "python binding table and header disagree"
    assert lib.rdhip_version() == 10000


def test_no_cpu_fallback_and_product_never_touches_oracle():
    pkg = os.path.join(ROOT, "robotdynamics.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".jl")) and "build" not in dp:
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "rd_oracle" not in txt and "oracle/" not in txt and "import oracle" not in txt, f"{f} references the oracle"
    if RD.device_count() == 0:
        with pytest.raises(RD.NoDeviceError):
            RD.Context(0)
        with pytest.raises(RD.NoDeviceError):
            RD.dynamics(RD.Cartpole(), np.zeros(4), np.zeros(1))


MODELS = [
    (lambda R: RD.Cartpole(), lambda r: O.cartpole(), False),
    (lambda R: RD.DoubleIntegrator(2), lambda r: O.double_integrator(2), False),
    (lambda R: RD.Satellite(R=R), lambda r: O.Model(O.SATELLITE, O.satellite().params, rot=r), True),
    (lambda R: RD.Quadrotor(R=R), lambda r: O.Model(O.QUADROTOR, O.quadrotor().params, rot=r), True),
    (lambda R: RD.SatellitePair(R=R), lambda r: O.Model(O.SATELLITE_PAIR, O.satellite_pair().params, rot=r), True),
    (lambda R: RD.LieSatellite(R=R), lambda r: O.Model(O.LIE_SATELLITE, [1.0, 1.0, 1.0], rot=r), True),
]


@pytest.mark.parametrize("mk,omk,lie", MODELS)
def test_dimension_queries_match_oracle_bit_exact(mk, omk, lie):
    for R, r in ((RD.QuatRotation, O.ROT_QUAT), (RD.MRP, O.ROT_MRP), (RD.RodriguesParam, O.ROT_RP)):
        m, om = mk(R), omk(r)
        assert RD.dims(m) == om.dims()
        assert (RD.state_dim(m), RD.control_dim(m), RD.output_dim(m), RD.errstate_dim(m)) == (om.n, om.m, om.p, om.e)
        assert RD.jacobian_width(m) == om.e + om.m and RD.input_dim(m) == om.n + om.m
        if lie:
            s = RD.LieState(m)
            assert (s.R.code, s.P) == om.liestate()
            assert isinstance(RD.statevectortype(m), RD.RotationState)
            dm = RD.DiscretizedDynamics(m, RD.RK4)       # trait forwarding, src/discretized_dynamics.jl:186-189
            assert RD.dims(dm) == om.dims() and RD.errstate_dim(dm) == om.e and isinstance(RD.statevectortype(dm), RD.RotationState)
        else:
            assert isinstance(RD.statevectortype(m), RD.EuclideanState)
            with pytest.raises(TypeError):
                RD.LieState(m)
        if not lie:
            break


def test_reference_index_tables():
    # test/liestate.jl:33-48
    s = RD.LieState(RD.QuatRotation, (3, 2, 3))
    assert len(s) == 16 and RD.errstate_dim(s) == 14
    assert RD.vec_inds(s, 1) == range(1, 4) == RD.inds(s, 1)
    assert RD.rot_inds(s, 1) == range(4, 8) == RD.inds(s, 2)
    assert RD.vec_inds(s, 2) == range(8, 10) == RD.inds(s, 3)
    assert RD.rot_inds(s, 2) == range(10, 14) == RD.inds(s, 4)
    assert RD.vec_inds(s, 3) == range(14, 17) == RD.inds(s, 5)
    assert RD.QuatState(16, (4, 10)) == s and RD.QuatState(16, [4, 10]) == s
    assert RD.num_rotations(s) == 2
    # test/liestate.jl:56-67 (RodriguesParam)
    s = RD.LieState(RD.RodriguesParam, (3, 2, 3))
    assert RD.LieState(RD.RodriguesParam, 3, 2, 3) == s
    assert len(s) == 14 and RD.errstate_dim(s) == 14
    assert RD.rot_inds(s, 1) == range(4, 7) and RD.vec_inds(s, 2) == range(7, 9)
    assert RD.rot_inds(s, 2) == range(9, 12) and RD.vec_inds(s, 3) == range(12, 15)
    # test/liemodel.jl:24-26
    m = RD.LieSatellite()
    assert RD.LieState(m) == RD.QuatState(7, (4,)) and RD.state_dim(m) == 7 and RD.errstate_dim(m) == 6
    # test/rigidbody_test.jl:60,69,88-98 ; test/rigid_body_jacobians.jl:5
    b = RD.Body()
    assert RD.dims(b) == (13, 6, 13)
    assert RD.LieState(b) == RD.LieState(RD.QuatRotation, (3, 6)) and RD.rotation_type(b) is RD.QuatRotation
    g = RD.gen_inds(b)
    assert g["q"] == range(4, 8) and g["u"] == range(14, 20) and g["r"] == range(1, 4) and g["v"] == range(8, 11) and g["w"] == range(11, 14)
    g = RD.gen_inds(RD.Body(RD.MRP))
    assert g["q"] == range(4, 7) and g["v"] == range(7, 10) and g["u"] == range(13, 19)
    for R in (RD.QuatRotation, RD.MRP, RD.RodriguesParam):      # test/rigidbody_test.jl:79-85
        assert RD.state_dim(RD.Body(R)) == 9 + RD.params(R)
    assert RD.params(RD.QuatRotation) == 4 and RD.params(RD.MRP) == 3 and RD.params(RD.RodriguesParam) == 3
    x0, u0 = RD.zeros(b)                                         # test/rigidbody_test.jl:72-76
    assert list(x0[3:7]) == [1, 0, 0, 0] and not u0.any()
    x, u = RD.rand(b, np.random.default_rng(0))
    assert abs(np.linalg.norm(x[3:7]) - 1) < 1e-14 and len(x) == 13 and len(u) == 6
    x, u = RD.rand(RD.SatellitePair(), np.random.default_rng(0))  # examples/satellite_pair.jl:46-48
    assert abs(np.linalg.norm(x[0:4]) - 1) < 1e-14 and abs(np.linalg.norm(x[7:11]) - 1) < 1e-14


def test_default_traits():
    c = RD.Cartpole()
    assert RD.default_diffmethod(c) == RD.UserDefined() and RD.default_signature(c) == RD.StaticReturn()
    assert RD.default_diffmethod(RD.DiscretizedDynamics(c, RD.RK4)) == RD.ForwardAD()   # src/discretized_dynamics.jl:190
    assert RD.velocity_frame(RD.Quadrotor(bodyframe=True)) == "body" and RD.velocity_frame(RD.Quadrotor()) == "world"
    with pytest.raises(ValueError):
        RD.ImplicitFunctionTheorem(RD.ImplicitFunctionTheorem(RD.ForwardAD()))
    assert RD.functioninputs(c) == RD.StateControl()


def test_knotpoint_semantics():
    # test/knotpoints.jl: data views, terminal control = zeros
    x, u = np.arange(1.0, 5.0), np.array([7.0])
    z = RD.KnotPoint(x, u, 0.5, 0.1)
    assert RD.dims(z) == (4, 1) and len(z) == 5 and RD.state_dim(z) == 4 and RD.control_dim(z) == 1
    assert list(RD.state(z)) == [1, 2, 3, 4] and list(RD.control(z)) == [7]
    assert RD.time(z) == 0.5 and RD.timestep(z) == 0.1 and not RD.is_terminal(z)
    RD.state(z)[0] = 9.0                       # state(z) is a view (src/knotpoint.jl:48)
    assert z[0] == 9.0
    zt = RD.KnotPoint(4, 1, z.z.copy(), 1.0, 0.0)
    assert RD.is_terminal(zt) and list(RD.control(zt)) == [0.0] and zt.z[4] == 7.0     # src/knotpoint.jl:67,123
    z2 = RD.StaticKnotPoint(z, np.zeros(5))     # src/knotpoint.jl:225-227
    assert z2.t == z.t and z2.dt == z.dt and not z2.z.any()
    with pytest.raises(AssertionError):
        RD.KnotPoint(3, 1, np.zeros(5), 0.0, 0.1)
    with pytest.raises(TypeError):
        RD.KnotPoint(x, u, 0.1)                 # v0.3 3-arg constructor does not exist in v0.4.8
    RD.setstate_(z, np.zeros(4)); RD.setcontrol_(z, [2.0]); RD.settime_(z, 3.0); RD.settimestep_(z, 0.0)
    assert not RD.state(z).any() and z.z[4] == 2.0 and z.t == 3.0 and RD.is_terminal(z)


def test_packed_trajectory_time_grids():
    # src/trajectories.jl:92-109, 355-400 ; test/trajectories.jl
    N, n, m = 11, 4, 2
    X, U = np.random.default_rng(0).random((N, n)), np.random.default_rng(1).random((N - 1, m))
    Z = RD.PackedTrajectory.from_states_controls(X, U, dt=0.1)
    assert len(Z) == N and RD.dims(Z) == ([n] * N, [m] * N, N)
    np.testing.assert_allclose(RD.gettimes(Z), np.arange(N) * 0.1)
    assert Z.dt[-1] == 0.0 and np.allclose(Z.dt[:-1], 0.1) and RD.is_terminal(Z[N - 1]) and not RD.has_terminal_control(Z)
    assert not RD.controls(Z)[-1].any()
    np.testing.assert_array_equal(RD.states(Z), X)
    Z2 = RD.PackedTrajectory.from_states_controls(X, np.vstack([U, U[-1:]]), tf=2.0)
    assert Z2.t[-1] == 2.0 and np.allclose(Z2.dt[:-1], 0.2) and np.isinf(Z2.dt[-1]) and RD.has_terminal_control(Z2)
    z3 = Z[3]
    z3.z[0] = 42.0                               # Z[k] is a view into the packed data
    assert Z.data[3, 0] == 42.0 and z3.t == pytest.approx(0.3)
    with pytest.raises(ValueError):
        RD._gettimeinfo(N=5)
    np.testing.assert_allclose(RD._gettimeinfo(dt=[0.1, 0.2, 0.3]), [0, 0.1, 0.3, 0.6])
    with pytest.raises(ValueError):
        RD.PackedTrajectory(4, 2, np.zeros((5, 7)), 0.0, 0.1)


def test_layout_checks_and_dynamics_jacobian():
    m = RD.Cartpole()
    J = RD.alloc_jacobians(m, 3)
    assert J.shape == (3, 4, 5) and J.base.shape == (3, 5, 4) and J.base.flags.c_contiguous
    D = RD.DynamicsJacobian(m)                    # test/jacobian_test.jl
    assert D.shape == (4, 5) and D.A.shape == (4, 4) and D.B.shape == (4, 1) and D.data.flags.f_contiguous
    D.A[1, 2] = 5.0
    assert D[1, 2] == 5.0 and RD.get_A(D) is D.A and RD.get_B(D) is D.B
    D2 = RD.DynamicsJacobian(np.arange(20.0).reshape(4, 5))
    assert D2.B[0, 0] == 4.0


def _gloo_worker(rank, world, port, N, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "robotdynamics.jl_amd"))
    sys.path.insert(0, ROOT)
    from rdhip import parallel as P
    from oracle import rd_oracle as O2
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    Z = rng.random((N, 5))
    lo, hi = P.shard_range(N, rank, world)
    # the per-rank kernel is stood in for by the oracle here (no GPU in this test); what is under test is the
    # sharding + exchange logic of rdhip.parallel
    Jl, _ = O2.cartpole().jacobian(O2.RK4, O2.FORWARD_AD, Z[lo:hi], 0.0, 0.1) if hi > lo else (np.zeros((0, 4, 5)), None)
    local = torch.as_tensor(np.ascontiguousarray(Jl))
    full = P.gather_stacked(local, N, "all")
    root = P.gather_stacked(local, N, "root")
    none = P.gather_stacked(local, N, "none")
    ref, _ = O2.cartpole().jacobian(O2.RK4, O2.FORWARD_AD, Z, 0.0, 0.1)
    ok = bool(np.array_equal(full.numpy(), ref)) and none is local
    ok = ok and ((root is None) if rank != 0 else bool(np.array_equal(root.numpy(), ref)))
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("N", [10, 7])
def test_world_size_2_sharding_and_gather_gloo(N):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + N) % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, N, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _stub_comm_worker(rank, world, port, stub, logdir, sizes, q):
    """one rank of a 2-process communicator on a CPU-only host: torch.distributed (gloo) carries the RCCL id, the library's exchange runs
    against the recording stand-in, on contexts without a device (RDHIP_DRY_RUN)"""
    import ctypes as C
    import os as _os
    _os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                        "RDHIP_RCCL_LIB": stub, "RDHIP_DRY_RUN": "1", "RDHIP_STUB_LOG": _os.path.join(logdir, f"rank{rank}.log")})
    import torch.distributed as dist
    from rdhip import parallel as P
    from rdhip import _lib as L
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = P.Comm.from_torch_distributed(rank)           # rank 0 draws the id (stub), broadcast over gloo, rdhip_comm_create_rank on every rank
        nr, nl = C.c_int(), C.c_int()
        L.check(L.lib().rdhip_comm_size(comm.handle, C.byref(nr), C.byref(nl)))
        shard = np.full(max(sizes[rank], 1), rank + 1, dtype=np.uint8)
        stack = np.zeros(sum(sizes) + 1, dtype=np.uint8)

        class Buf:                                           # what Comm.gather needs of a device tensor
            def __init__(self, a): self.a = a
            def data_ptr(self): return self.a.ctypes.data
        out = {"size": (nr.value, nl.value), "shard": shard.ctypes.data, "stack": stack.ctypes.data}
        for mode in ("root", "all"):
            stack[:] = 0
            open(_os.environ["RDHIP_STUB_LOG"], "a").write(f"mode {mode}\n")
            comm.gather(mode, [Buf(shard)], sizes, [Buf(stack)], root=1)
            out[mode + "_own"] = bytes(stack).count(bytes([rank + 1]))
        comm.close()
        # a compute entry point on the device-less context must refuse, not crash
        ctx = L.Context(rank)
        rc = L.lib().rdhip_ctx_synchronize(ctx.handle)
        out["sync_rc"] = rc
        q.put((rank, out))
    except Exception as e:                                   # noqa: BLE001
        q.put((rank, {"error": f"{type(e).__name__}: {e}"}))
    dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [[4096, 4096], [1000, 3000], [0, 77]])
def test_native_exchange_bookkeeping_world_size_2_against_a_recording_rccl(sizes, tmp_path):
    """The RCCL exchange of the C ABI (csrc/comm.hip: rdhip_comm_unique_id / rdhip_comm_create_rank / rdhip_gather) with TWO ranks, on a host without
    a GPU: a recording stand-in for librccl (tests/stubs/rccl_stub.c) is loaded through RDHIP_RCCL_LIB and every rank's calls are compared with
    the tables the exchange must produce -- ROOT: every non-root rank sends its shard to the root, the root receives each at its offset in the
    stack inside one group; ALL: ncclAllGather for equal shards, the grouped send / recv pattern for ragged ones; a rank's own shard is a local
    copy at its offset; empty shards produce no traffic."""
    import subprocess
    import torch.multiprocessing as mp
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", stub, os.path.join(ROOT, "tests", "stubs", "rccl_stub.c")])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + sum(sizes)) % 2000
    procs = [ctx.Process(target=_stub_comm_worker, args=(r, 2, port, stub, str(tmp_path), sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert all("error" not in v for v in res.values()), res
    off = [0, sizes[0]]
    for rank in range(2):
        info = res[rank]
        assert info["size"] == (2, 1) and info["sync_rc"] == -5          # RDHIP_ERR_NO_DEVICE
        lines = open(tmp_path / f"rank{rank}.log").read().split("\n")
        assert any(ln.startswith(f"init rank={rank} nranks=2 idok=1") for ln in lines)
        calls = {"root": [], "all": []}
        mode = None
        for ln in lines:
            if ln.startswith("mode "):
                mode = ln.split()[1]
            elif mode and ln.split(" ")[0] in ("send", "recv", "allgather", "localcopy", "groupstart", "groupend"):
                calls[mode].append(ln)
        other = 1 - rank

        def parse(ln):
            f = dict(tok.split("=") for tok in ln.split()[1:] if "=" in tok)
            return ln.split()[0], {k: int(v) for k, v in f.items()}
        # ---- ROOT (root = rank 1)
        ops = [parse(ln) for ln in calls["root"]]
        sends = [o for o in ops if o[0] == "send"]; recvs = [o for o in ops if o[0] == "recv"]; copies = [o for o in ops if o[0] == "localcopy"]
        if rank == 1:
            assert not sends
            assert [(r["peer"], r["bytes"], r["buf"] - info["stack"]) for _, r in recvs] == ([(0, sizes[0], 0)] if sizes[0] else [])
            assert [(c["bytes"], c["dst"] - info["stack"], c["src"] - info["shard"]) for _, c in copies] == ([(sizes[1], off[1], 0)] if sizes[1] else [])
            assert info["root_own"] >= sizes[1]
        else:
            assert [(s_["peer"], s_["bytes"], s_["buf"] - info["shard"]) for _, s_ in sends] == ([(1, sizes[0], 0)] if sizes[0] else [])
            assert not recvs and not copies
        assert [o[0] for o in ops].count("groupstart") == 1 and [o[0] for o in ops].count("groupend") == 1
        # ---- ALL
        ops = [parse(ln) for ln in calls["all"]]
        if sizes[0] == sizes[1]:
            ag = [o for o in ops if o[0] == "allgather"]
            assert [(a["bytes"], a["send"] - info["shard"], a["recv"] - info["stack"]) for _, a in ag] == [(sizes[0], 0, 0)]
            assert not [o for o in ops if o[0] in ("send", "recv", "localcopy")]
        else:
            sends = [o for o in ops if o[0] == "send"]; recvs = [o for o in ops if o[0] == "recv"]; copies = [o for o in ops if o[0] == "localcopy"]
            assert [(s_["peer"], s_["bytes"]) for _, s_ in sends] == ([(other, sizes[rank])] if sizes[rank] else [])
            assert [(r["peer"], r["bytes"], r["buf"] - info["stack"]) for _, r in recvs] == ([(other, sizes[other], off[other])] if sizes[other] else [])
            assert [(c["bytes"], c["dst"] - info["stack"]) for _, c in copies] == ([(sizes[rank], off[rank])] if sizes[rank] else [])


def test_shard_ranges_cover_everything():
    from rdhip import parallel as P
    for N in (0, 1, 7, 64, 65536, 262144 + 3):
        for world in (1, 2, 4, 8):
            spans = [P.shard_range(N, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert sum(P.shard_sizes(N, world)) == N





def test_custom_model_jit_compiles_without_a_gpu():
    """hipRTC plug-in (user-defined dynamics): creation, integer queries and JIT compilation need no GPU; a body that
    does not compile raises ValueError carrying the compiler log."""
    m = RD.CustomModel(4, 1, H.CARTPOLE_SRC, params=(1.0, 0.2, 0.5, 9.81))
    assert RD.dims(m) == (4, 1, 4) and RD.errstate_dim(m) == 4 and isinstance(RD.statevectortype(m), RD.EuclideanState)
    m.precompile("ForwardAD", RD.RK4, np.float64)
    m.precompile("evaluate", None, np.float32)
    bad = RD.CustomModel(2, 1, "xdot[0] = x[1]; xdot[1] = undefined_symbol;")
    with pytest.raises(ValueError, match="undefined_symbol"):
        bad.precompile("evaluate", RD.RK4)
    lie = RD.CustomModel(7, 3, "xdot[0] = u[0];", liestate=(RD.QuatRotation, (3, 0)))
    assert RD.errstate_dim(lie) == 6 and isinstance(RD.statevectortype(lie), RD.RotationState)
    lie.precompile("error_expansion", RD.RK4)           # the fused error-state expansion kernel: quaternion at state index 3 of LieState(R,(3,0))
    pair = RD.CustomModel(14, 6, "for (int i = 0; i < 14; ++i) xdot[i] = x[i] * u[i % 6];", liestate=(RD.QuatRotation, (0, 3, 3)))
    pair.precompile("error_expansion", RD.RK3, np.float32)   # two rotations (state indices 0 and 7)
    with pytest.raises(ValueError):
        RD.dims(RD.CustomModel(8, 3, "xdot[0] = u[0];", liestate=(RD.QuatRotation, (3, 0))))     # LieState length 7 != n
    # time-dependent dynamics: `t` is available in the body (dynamics(model, x, u, t)); a body written for the time-invariant signature that
    # declares its own `t` still compiles (it shadows the argument); every kernel family takes the time through
    tv = RD.CustomModel(2, 1, H.TV_PENDULUM_SRC, params=(9.81, 2.0, 3.0, 0.1))
    for what, rule in (("evaluate", RD.RK4), ("ForwardAD", RD.RK3), ("UserDefined", RD.RK4), ("FiniteDifference", RD.RK2), ("rollout", RD.RK4)):
        tv.precompile(what, rule)
    for op in (7, 8, 9, 10, 13, 14, 15):                                        # the ImplicitMidpoint kernels (LDS and register forms)
        RD.lib().rdhip_model_precompile(tv._handle, op, 4, 0) == 0 or pytest.fail(RD.lib().rdhip_last_error(None).decode())
    tvs = RD.CustomModel(13, 6, H.TV_SATELLITE_SRC, params=(2.0, 2.0, 3.0, 1.0, 0.7, 0.4, 2.5), liestate=(RD.QuatRotation, (3, 6)))
    tvs.precompile("ForwardAD", RD.RK4)
    tvs.precompile("error_expansion", RD.RK4)
    tvs.precompile("implicit_jacobian", RD.ImplicitMidpoint)                     # the IFT Jacobian as a split-role tile kernel (n + m > 10)
    # a declared block structure (rdhip_model_set_cascade): the cascade forms of FiniteDifference, rollout and the Newton solve compile without a GPU
    tvs.set_cascade([2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 0, 0, 0], [2, 2, 2, 0, 0, 0])      # w | q | r, v ; thrust enters r, v ; torque enters w
    tvs.precompile("FiniteDifference", RD.RK4)
    tvs.precompile("rollout", RD.RK3, np.float32)
    tvs.precompile("implicit_solve", RD.ImplicitMidpoint)
    assert RD.lib().rdhip_model_precompile(tvs._handle, 17, 4, 0) == 0          # the implicit rollout along the declared cascade
    # a rigid body defined the reference's way -- `forces` and `moments` (src/rigidbody.jl:244-258) -- as two snippets of HIP source
    crb = RD.CustomRigidBody(6, "quat_rotate(q, u, F);", "tau[0] = u[3]; tau[1] = u[4]; tau[2] = u[5] * P[0];", mass=2.0, J=np.diag([2.0, 3.0, 1.0]), params=(1.0,),
                             bodyframe=True, force_only_controls=(0, 1, 2))
    assert RD.dims(crb)[:2] == (13, 6) and RD.errstate_dim(crb) == 12 and crb.cascade[1] == (2, 2, 2, 0, 0, 0)
    crb.precompile("ForwardAD", RD.RK4)
    crb.precompile("rollout", RD.RK4)
    for bad in (([0] * 12, [0] * 6), ([0] * 13, [7] * 6), ([0, 0, 0, 2, 2, 2, 2, 0, 0, 0, 0, 0, 0], [0] * 6)):     # wrong length / block out of range / empty block 1
        with pytest.raises(ValueError):
            tvs.set_cascade(*bad)
    RD.CustomModel(2, 1, "const T t = p[0]; xdot[0] = x[1] * t; xdot[1] = u[0];", params=(2.0,)).precompile("evaluate", RD.RK4)
    with pytest.raises(NotImplementedError):                                      # LieDesc tables hold 4 rotations
        RD.dims(RD.CustomModel(16, 1, "xdot[0] = u[0];", liestate=(RD.MRP, (0, 0, 0, 0, 0, 1))))


def test_jit_disk_cache_is_opt_in_and_survives_corruption(tmp_path, monkeypatch):
    """RDHIP_CACHE_DIR: compiled kernels of custom models are stored as code objects keyed by a hash of the generated source, the kernel
    templates and the HIP version; a second model object (a new process in real life) loads them instead of compiling; a damaged file is
    ignored and rewritten."""
    import time
    monkeypatch.setenv("RDHIP_CACHE_DIR", str(tmp_path))
    src = H.CARTPOLE_SRC + "\n// cache test " + str(time.time())         # unique source -> unique key
    t0 = time.perf_counter()
    RD.CustomModel(4, 1, src, params=(1.0, 0.2, 0.5, 9.81)).precompile("ForwardAD", RD.RK4)
    cold = time.perf_counter() - t0
    files = sorted(tmp_path.glob("rdhip_*.co"))
    assert len(files) == 1 and files[0].stat().st_size > 1000
    t0 = time.perf_counter()
    RD.CustomModel(4, 1, src, params=(1.0, 0.2, 0.5, 9.81)).precompile("ForwardAD", RD.RK4)
    warm = time.perf_counter() - t0
    assert warm < 0.25 * cold, (cold, warm)
    files[0].write_bytes(b"garbage")
    RD.CustomModel(4, 1, src, params=(1.0, 0.2, 0.5, 9.81)).precompile("ForwardAD", RD.RK4)     # recompiles and rewrites
    assert files[0].stat().st_size > 1000
    monkeypatch.delenv("RDHIP_CACHE_DIR")
    RD.CustomModel(4, 1, src + " ", params=(1.0, 0.2, 0.5, 9.81)).precompile("evaluate", RD.RK4)
    assert len(list(tmp_path.glob("rdhip_*.co"))) == 1                   # no cache without the variable


def test_header_is_plain_c_and_the_c_demo_links_and_runs():
    """include/rdhip.h must be consumable by a C compiler (the boundary is a C ABI): compile examples/c_abi_demo.c as C99 with
    -Wall -Wextra -Werror, link it against librdhip.so and run it (on a box without a GPU it exercises the model queries and exits 0)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_build", "c_abi_demo")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.dirname(RD.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_abi_demo.c"), "-L", libdir, "-lrdhip", "-lm", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "n=13 m=6 p=13 errstate_dim=12" in out.stdout


def test_every_documented_tuning_knob_is_accepted_and_nothing_else():
    """include/rdhip.h lists the keys of RDHIP_TUNE / rdhip_tune(): every key of that list must parse (the parser and the header are kept in step by this
    test), an unknown key is an error that names it, and "reset" restores the environment's settings.  Host-only: no launch is issued."""
    import rdhip._lib as L
    text = open(L.HEADER_PATH).read()
    block = text[text.index("Keys (csrc/common.hpp TuneKnobs):"):text.index("Not tuning, and still their own environment variables")]
    keys = sorted(set(re.findall(r"\b([a-z_]+[a-z0-9_]*)=", block)))
    assert {"static_grid", "dyn_grid", "dynamic_tiles", "small", "small_max", "rollout", "rolljac", "fused_ratio", "fused_jac_workgroups", "fused_tj_max_knots",
            "rolljac_pipe_min_knots", "multi_main", "fused_two_per_cu", "mixed", "mixed_idle", "ift"} <= set(keys), keys
    src = open(os.path.join(ROOT, "robotdynamics.jl_amd", "csrc", "api.hip")).read()
    parsed = set(re.findall(r'key == "([a-z_0-9]+)"', src))
    assert parsed == set(keys), (sorted(parsed - set(keys)), sorted(set(keys) - parsed))        # header and parser name the same knobs
    try:
        for k in keys:
            RD.tune(f"{k}=1")
        with pytest.raises(Exception, match="no_such_knob"):
            RD.tune("no_such_knob=1")
    finally:
        RD.tune("reset")
