#!/usr/bin/env python3
"""tools/bench_configs.py -- single-GPU timing of every BASELINE.json config and of the secondary kernels, plus the N-sweep
that shows the launch-bound -> bandwidth-bound transition.  One JSON line per measurement:
  {"name", "N", "dtype", "us_per_launch", "units_per_s", "alg_bytes", "GBps", "frac_of_8TBps"}
Kernel time = one HIP-event pair around `reps` back-to-back launches on torch's current stream (what rdhip borrows).
Outputs rotate over enough buffers to exceed the 256 MiB Infinity Cache when they are large."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import rdhip as RD  # noqa: E402

DEV = "cuda:0"


def rigid_inputs(N, m, rng, lo=0.0, hi=1.0):
    q = rng.standard_normal((N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    u = lo + (hi - lo) * rng.random((N, m))
    return np.concatenate([rng.random((N, 3)), q, rng.random((N, 6)), u], axis=1)


PREWARM_S = 0.12     # sustained load before every timed loop: clocks settle over ~100 ms of load; a handful of warm-up launches measures the
                     # ramp (3-7 % slower per launch: tools/abi_bench round 0 vs round 1)


def timed(fn, reps, warm=5):
    import time
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0, i = time.perf_counter(), 0
    while time.perf_counter() - t0 < PREWARM_S:
        for _ in range(20):
            fn(i); i += 1
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps    # us


OVCTX = None       # --overlap: an rdhip context that owns its stream, launch overlap on (rdhip_ctx_set_overlap); jac_case issues its calls on it


def timed_ctx(fn, reps, ctx, warm=5):
    """timed() for calls issued on an rdhip context's own stream(s): the library's stopwatch (events on the context's stream behind a join)."""
    import time
    for i in range(warm):
        fn(i)
    ctx.synchronize()
    t0, i = time.perf_counter(), 0
    while time.perf_counter() - t0 < PREWARM_S:
        for _ in range(20):
            fn(i); i += 1
        ctx.synchronize()
    ctx.timer_start()
    for i in range(reps):
        fn(i)
    return ctx.timer_stop() * 1e3 / reps


def bare(thunks):
    """Run each checked Python call once while recording the C-ABI calls it makes, and return replays of just those C calls: the timed
    loops then measure the library (what a `ccall` host pays per call), not the ~13 us of Python argument checking per call, which is longer
    than most of the small-batch kernels."""
    from rdhip import _lib as L
    real = L.lib()

    class Proxy:
        def __init__(self):
            self.calls = []

        def __getattr__(self, name):
            f = getattr(real, name)

            def w(*a):
                self.calls.append((f, a))
                return f(*a)
            return w
    out = []
    for th in thunks:
        px = Proxy()
        L._lib = px
        try:
            th()
        finally:
            L._lib = real
        keep = ("rdhip_jacobian", "rdhip_error_expansion", "rdhip_errstate_jacobian", "rdhip_state_diff", "rdhip_nabla_errstate_jacobian",
                "rdhip_dynamics", "rdhip_discrete_dynamics", "rdhip_dynamics_error", "rdhip_dynamics_error_jacobian")
        launches = [(f, a) for f, a in px.calls if f.__name__ in keep]
        assert launches, "no batched C call recorded"
        out.append(lambda launches=launches: [f(*a) for f, a in launches])
    return out


def emit(name, N, dtype, us, alg_bytes, extra=None):
    out = {"name": name, "N": N, "dtype": dtype, "us_per_launch": round(us, 3), "units_per_s": N / (us * 1e-6),
           "alg_bytes": alg_bytes, "GBps": alg_bytes / (us * 1e-6) / 1e9, "frac_of_8TBps": alg_bytes / (us * 1e-6) / 8e12}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


def nbuf_for(bytes_per_buf):
    nb = max(2, min(8, int(600e6 // max(bytes_per_buf, 1)) + 1))
    # launch overlap alternates two streams: with an odd rotation a buffer changes streams every time round, and the write-after-write edge that
    # (correctly) follows serialises every launch -- an artefact of the rotation, not of a caller's loop
    return nb + (nb & 1) if OVCTX is not None else nb


def jac_case(name, model, Zn, rule, diff, dtype, reps, with_y=True):
    N = Zn.shape[0]
    n, m = RD.state_dim(model), RD.control_dim(model)
    T = 8 if dtype == np.float64 else 4
    fun = RD.DiscretizedDynamics(model, rule) if rule is not None else model
    nb = nbuf_for(N * n * (n + m) * T)
    # the INPUTS rotate with the outputs (separate Z / t / dt arrays per buffer): one input batch re-read by every launch stays in the 256 MiB Infinity
    # Cache up to ~1 M satellite knot points and flatters exactly those sizes (profiles/r04/bign_tile_order.txt: 0.745 -> 0.646 at 1 M)
    Zs = [RD.PackedTrajectory(n, m, Zn.astype(dtype), np.arange(N) * 0.1, np.full(N, 0.1)).to(DEV) for _ in range(nb)]
    Z = Zs[0]
    Js = [RD.alloc_jacobians(fun, N, Z.data) for _ in range(nb)]
    Ys = [torch.empty((N, n), dtype=Z.data.dtype, device=DEV) for _ in range(nb)]
    sig = RD.StaticReturn()
    calls = [RD.prepare_jacobian(sig, diff, fun, Js[b], Ys[b] if with_y else None, Zs[b], ctx=OVCTX) for b in range(nb)]
    torch.cuda.synchronize()
    us = timed_ctx(lambda i: calls[i % nb](), reps, OVCTX) if OVCTX is not None else timed(lambda i: calls[i % nb](), reps)
    alg = N * ((n + m) * T + 16 + n * (n + m) * T + (n * T if with_y else 0))
    emit(name, N, "f64" if T == 8 else "f32", us, alg, {"output_buffers": nb, "input_buffers": nb, "launch_overlap": OVCTX is not None})


def errexp_case(name, model, Zn, rule, dtype, reps):
    N = Zn.shape[0]
    n, m, e = RD.state_dim(model), RD.control_dim(model), RD.errstate_dim(model)
    T = 8 if dtype == np.float64 else 4
    fun = RD.DiscretizedDynamics(model, rule)
    Z = RD.PackedTrajectory(n, m, Zn.astype(dtype), np.arange(N) * 0.1, np.full(N, 0.1)).to(DEV)
    nb = nbuf_for(N * e * (e + m) * T)
    Es = [RD.alloc_error_expansions(fun, N, Z.data) for _ in range(nb)]
    Ys = [torch.empty((N - 1, n), dtype=Z.data.dtype, device=DEV) for _ in range(nb)]
    calls = bare([lambda b=b: RD.error_expansion_(fun, Es[b], Ys[b], Z) for b in range(nb)])
    us = timed(lambda i: calls[i % nb](), reps)
    alg = (N - 1) * ((n + m) * T + 16 + e * (e + m) * T + n * T)
    emit(name, N - 1, "f64" if T == 8 else "f32", us, alg, {"output_buffers": nb})


def eval_case(name, model, Zn, rule, dtype, reps):
    N = Zn.shape[0]
    n, m = RD.state_dim(model), RD.control_dim(model)
    T = 8 if dtype == np.float64 else 4
    fun = RD.DiscretizedDynamics(model, rule) if rule is not None else model
    Z = RD.PackedTrajectory(n, m, Zn.astype(dtype), np.arange(N) * 0.1, np.full(N, 0.1)).to(DEV)
    nb = nbuf_for(N * n * T)
    Ys = [torch.empty((N, n), dtype=Z.data.dtype, device=DEV) for _ in range(nb)]
    RD.evaluate_(fun, Ys[0], Z)          # the checked call once; then the bare C call (what a ccall host pays), so that the 13 us of
    from rdhip import _lib as L           # Python argument checking do not hide a ~8 us kernel
    ctx = L.context_for(0, torch.cuda.current_stream(0).cuda_stream)
    lib, F = L.lib(), (L.F64 if T == 8 else L.F32)
    cm = model._handle
    zp, tp, dp = Z.data.data_ptr(), Z.t.data_ptr(), Z.dt.data_ptr()
    ys = [y.data_ptr() for y in Ys]
    if rule is None:
        call = lambda i: lib.rdhip_dynamics(ctx.handle, cm, F, N, zp, tp, dp, ys[i % nb], L.DEVICE)     # noqa: E731
    else:
        rc = RD.core._rule_code(rule() if isinstance(rule, type) else rule)
        call = lambda i: lib.rdhip_discrete_dynamics(ctx.handle, cm, rc, F, N, zp, tp, dp, ys[i % nb], L.DEVICE)     # noqa: E731
    us = timed(call, reps)
    emit(name, N, "f64" if T == 8 else "f32", us, N * ((n + m) * T + 16 + n * T), {"output_buffers": nb})


def errstate_case(name, model, Zn, reps, clear):
    N = Zn.shape[0]
    n, m, e = RD.state_dim(model), RD.control_dim(model), RD.errstate_dim(model)
    Z = RD.PackedTrajectory(n, m, Zn, np.arange(N) * 0.1, np.full(N, 0.1)).to(DEV)
    nb = nbuf_for(N * n * e * 8)
    Gs = [RD.alloc_errstate_jacobians(model, N, Z.data) for _ in range(nb)]
    calls = bare([lambda b=b: RD.errstate_jacobian_(model, Gs[b], Z, clear=clear) for b in range(nb)])
    us = timed(lambda i: calls[i % nb](), reps)
    alg = N * (n * 8 + (n * e * 8 if clear else (n - 4 + 12) * 8))
    emit(name, N, "f64", us, alg, {"output_buffers": nb, "clear": clear})


def statediff_case(name, model, Zn, reps, rng):
    N = Zn.shape[0]
    n, e = RD.state_dim(model), RD.errstate_dim(model)
    X = torch.from_numpy(np.ascontiguousarray(Zn[:, :n])).to(DEV)
    X0 = torch.from_numpy(np.ascontiguousarray(Zn[::-1, :n].copy())).to(DEV)
    dX = torch.empty((N, e), dtype=torch.float64, device=DEV)
    call = bare([lambda: RD.state_diff_(model, dX, X, X0)])[0]
    us = timed(lambda i: call(), reps)
    emit(name, N, "f64", us, N * (2 * n * 8 + e * 8))
    dG = torch.empty((N, e, e), dtype=torch.float64, device=DEV).transpose(1, 2)
    call = bare([lambda: RD.nabla_errstate_jacobian_(model, dG, X, X0, clear=True)])[0]
    us = timed(lambda i: call(), reps)
    emit(name.replace("state_diff", "nabla_errstate_jacobian clear=True"), N, "f64", us, N * (2 * n * 8 + e * e * 8))


def rollout_case(name, model, M, K, rule, dtype, reps, rng, tm=False):
    n, m = RD.state_dim(model), RD.control_dim(model)
    T = 8 if dtype == np.float64 else 4
    Zn = (rigid_inputs(M * K, m, rng) if n == 13 else rng.random((M * K, n + m))).astype(dtype)
    shape = (K, M) if tm else (M, K)
    data = torch.from_numpy(Zn.reshape(*shape, n + m)).to(DEV)
    tk = np.arange(K) * 0.01
    t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[:, None] if tm else tk[None, :], shape))).to(DEV)
    dt = torch.full(shape, 0.01, dtype=torch.float64, device=DEV)
    fun = RD.DiscretizedDynamics(model, rule)
    us = timed(lambda i: RD.rollout_(RD.StaticReturn(), fun, {"data": data, "t": t, "dt": dt}, time_major=tm), reps, warm=2)
    emit(name + (" (time-major)" if tm else ""), M * K, "f64" if T == 8 else "f32", us, M * K * ((n + m) * T + 16 + n * T), {"trajectories": M, "knots": K})


def rollout_split_case(name, model, M, K, rule, dtype, reps, rng, tm=True):
    """rdhip_rollout_split (states and controls in arrays of their own) beside rdhip_rollout on the packed stack, same inputs, same box: algorithmic bytes =
    read u, t, dt; write x"""
    n, m = RD.state_dim(model), RD.control_dim(model)
    T = 8 if dtype == np.float64 else 4
    Zn = (rigid_inputs(M * K, m, rng) if n == 13 else rng.random((M * K, n + m))).astype(dtype)
    shape = (K, M) if tm else (M, K)
    data = torch.from_numpy(Zn.reshape(*shape, n + m)).to(DEV)
    tk = np.arange(K) * 0.01
    t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[:, None] if tm else tk[None, :], shape))).to(DEV)
    dt = torch.full(shape, 0.01, dtype=torch.float64, device=DEV)
    fun = RD.DiscretizedDynamics(model, rule)
    X = torch.empty(shape + (n,), dtype=data.dtype, device=DEV)
    x0 = (data[0, :, :n] if tm else data[:, 0, :n]).contiguous()
    U = (data[:K - 1, :, n:] if tm else data[:, :K - 1, n:]).contiguous()
    alg = M * (K - 1) * (m * T + 16 + n * T) + M * n * T
    us = timed(lambda i: RD.rollout_split_(RD.StaticReturn(), fun, X, U, t, dt, x0=x0, time_major=tm), reps, warm=2)
    emit(name + " rollout_split" + (" (time-major)" if tm else " (trajectory-major)"), M * K, "f64" if T == 8 else "f32", us, alg, {"trajectories": M, "knots": K, "entry_point": "rdhip_rollout_split"})
    us = timed(lambda i: RD.rollout_(RD.StaticReturn(), fun, {"data": data, "t": t, "dt": dt}, time_major=tm), reps, warm=2)
    emit(name + " rollout on the packed stack, same algorithmic bytes" + (" (time-major)" if tm else " (trajectory-major)"), M * K, "f64" if T == 8 else "f32", us, alg,
         {"trajectories": M, "knots": K, "entry_point": "rdhip_rollout_layout"})


def rollout_jacobian_case(name, model, M, K, rule, dtype, reps, rng):
    """forward pass + linearisation: (a) rollout_ then jacobian_ (two calls), (b) rollout_jacobian_ (one call), trajectory- and time-major"""
    n, m = RD.state_dim(model), RD.control_dim(model)
    T = 8 if dtype == np.float64 else 4
    Zn = (rigid_inputs(M * K, m, rng) if n == 13 else rng.random((M * K, n + m))).astype(dtype)
    fun = RD.DiscretizedDynamics(model, rule)
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    alg = M * K * ((n + m) * T + 16 + n * T) + M * K * ((n + m) * T + 16 + n * (n + m) * T)
    for tm in (False, True):
        shape = (K, M) if tm else (M, K)
        data = torch.from_numpy(Zn.reshape(shape + (n + m,))).to(DEV)
        tk = np.arange(K) * 0.01
        t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[:, None] if tm else tk[None, :], shape))).to(DEV)
        dt = torch.full(shape, 0.01, dtype=torch.float64, device=DEV)
        stack = {"data": data, "t": t, "dt": dt}
        J = RD.alloc_jacobians(fun, M * K, data)
        if not tm:
            flat = RD.PackedTrajectory(n, m, data.reshape(M * K, n + m), t.reshape(-1), dt.reshape(-1))

            def two_calls(i):
                RD.rollout_(sig, fun, stack)
                RD.jacobian_(sig, AD, fun, J, None, flat)
            us = timed(two_calls, reps, warm=2)
            emit(name + ": rollout_ + jacobian_ (two calls)", M * K, "f64" if T == 8 else "f32", us, alg, {"trajectories": M, "knots": K})
        us = timed(lambda i: RD.rollout_jacobian_(sig, AD, fun, J, stack, time_major=tm), reps, warm=2)
        emit(name + (": rollout_jacobian_ time-major" if tm else ": rollout_jacobian_ trajectory-major"), M * K, "f64" if T == 8 else "f32", us, alg,
             {"trajectories": M, "knots": K, "kernel_launches_per_call": RD.rollout_jacobian_(sig, AD, fun, J, stack, time_major=tm)})


def c5_case(reps, rng, M=512):
    """BASELINE config 5 (M = 512: one GPU's share of the 8-GPU split; M = 4096: the whole workload on ONE GPU): M Cartpole + M Satellite
    trajectories x 256 knots, fp32, rollouts + all Jacobians:
    (a) four calls (rollout_, jacobian_ per model), (b) one rdhip_rollout_jacobian call per model, (c) ONE rdhip_rollout_jacobian_multi call."""
    K, dtype = 256, np.float32
    tag = "C5 share" if M == 512 else "C5 whole"
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    funs, stacks_tm, stacks_tj, Js, flats, alg = [], [], [], [], [], 0
    for model, mctl in ((RD.Cartpole(), 1), (RD.Satellite(), 6)):
        n = RD.state_dim(model)
        Zn = (rigid_inputs(M * K, mctl, rng) if n == 13 else rng.random((M * K, n + mctl))).astype(dtype)
        fun = RD.DiscretizedDynamics(model, RD.RK4)
        tk = np.arange(K) * 0.01
        tm = {"data": torch.from_numpy(Zn.reshape(K, M, n + mctl)).to(DEV), "t": torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[:, None], (K, M)))).to(DEV),
              "dt": torch.full((K, M), 0.01, dtype=torch.float64, device=DEV)}
        tj = {"data": torch.from_numpy(Zn.reshape(M, K, n + mctl)).to(DEV), "t": torch.from_numpy(np.ascontiguousarray(np.broadcast_to(tk[None, :], (M, K)))).to(DEV),
              "dt": torch.full((M, K), 0.01, dtype=torch.float64, device=DEV)}
        funs.append(fun); stacks_tm.append(tm); stacks_tj.append(tj); Js.append(RD.alloc_jacobians(fun, M * K, tm["data"]))
        flats.append(RD.PackedTrajectory(n, mctl, tj["data"].reshape(M * K, n + mctl), tj["t"].reshape(-1), tj["dt"].reshape(-1)))
        alg += M * K * ((n + mctl) * 4 + 16 + n * 4) + M * K * ((n + mctl) * 4 + 16 + n * (n + mctl) * 4)

    def four(i):
        for p in range(2):
            RD.rollout_(sig, funs[p], stacks_tj[p])
            RD.jacobian_(sig, AD, funs[p], Js[p], None, flats[p])
    us = timed(four, reps, warm=2)
    emit(f"{tag} ({M} cartpole + {M} satellite x 256, fp32): rollout_ + jacobian_ per model (four calls)", 2 * M * K, "f32", us, alg)

    def two(i):
        for p in range(2):
            RD.rollout_jacobian_(sig, AD, funs[p], Js[p], stacks_tm[p], time_major=True)
    us = timed(two, reps, warm=2)
    emit(f"{tag}: rollout_jacobian_ per model, time-major (two calls)", 2 * M * K, "f32", us, alg)
    call = RD.PreparedRolloutJacobianMulti(sig, AD, funs, Js, stacks_tm, None, time_major=True)
    us = timed(lambda i: call(), reps, warm=2)
    emit(f"{tag}: ONE call (rdhip_rollout_jacobian_multi), time-major", 2 * M * K, "f32", us, alg, {"trajectories": 2 * M, "knots": K})


def multi_case(name, model, gen, nbatch, per, rule, dtype, reps, singles=True, diff=None):
    """`nbatch` independent trajectories of `per` knot points (separate arrays): (a) one rdhip_jacobian call each, (b) ONE rdhip_jacobian_multi launch."""
    n, m = RD.state_dim(model), RD.control_dim(model)
    T = 8 if dtype == np.float64 else 4
    fun = RD.DiscretizedDynamics(model, rule)
    sig, AD = RD.StaticReturn(), (diff or RD.ForwardAD())
    Zs = [RD.PackedTrajectory(n, m, gen(per).astype(dtype), np.arange(per) * 0.1, np.full(per, 0.1)).to(DEV) for _ in range(nbatch)]
    nb = nbuf_for(nbatch * per * n * (n + m) * T)
    Js = [[RD.alloc_jacobians(fun, per, Zs[0].data) for _ in range(nbatch)] for _ in range(nb)]
    Ys = [[torch.empty((per, n), dtype=Zs[0].data.dtype, device=DEV) for _ in range(nbatch)] for _ in range(nb)]
    alg = nbatch * per * ((n + m) * T + 16 + n * (n + m) * T + n * T)
    if singles:
        one = [[RD.prepare_jacobian(sig, AD, fun, Js[k][b], Ys[k][b], Zs[b]) for b in range(nbatch)] for k in range(nb)]
        us = timed(lambda i: [c() for c in one[i % nb]], reps)
        emit(name + f": {nbatch} separate launches", nbatch * per, "f64" if T == 8 else "f32", us, alg, {"batches": nbatch, "knots_per_batch": per})
    fused = [RD.prepare_jacobian_multi(sig, AD, fun, Js[k], Ys[k], Zs) for k in range(nb)]
    us = timed(lambda i: fused[i % nb](), reps)
    emit(name + ": ONE call (rdhip_jacobian_multi)", nbatch * per, "f64" if T == 8 else "f32", us, alg,
         {"batches": nbatch, "knots_per_batch": per, "kernel_launches_per_call": fused[0].launches})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true", help="also run the N sweep 2^12 .. 2^22 for C2/C3")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--only", default="", help="run only the rollout+Jacobian cases (value: rolljac)")
    ap.add_argument("--overlap", action="store_true", help="Jacobian cases on one rdhip context with launch overlap on (rdhip_ctx_set_overlap)")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    if a.overlap:
        global OVCTX
        from rdhip import _lib as L
        OVCTX = L.Context(0)
        OVCTX.set_overlap(True)
    rng = np.random.default_rng(7)
    f64, f32 = np.float64, np.float32
    sat, cart, quad = RD.Satellite(), RD.Cartpole(), RD.Quadrotor()
    AD, FD, UD = RD.ForwardAD(), RD.FiniteDifference(), RD.UserDefined()
    if a.only == "sweep":
        for p in range(12, 23):
            N = 1 << p
            jac_case("sweep satellite RK4 jacobian ForwardAD", sat, rigid_inputs(N, 6, rng), RD.RK4, AD, f64, max(5, min(a.reps, (1 << 24) // N)))
        for p in range(12, 23, 2):
            N = 1 << p
            jac_case("sweep cartpole RK4 jacobian ForwardAD", cart, rng.random((N, 5)), RD.RK4, AD, f64, max(5, min(a.reps, (1 << 25) // N)))
        jac_case("C5 cartpole RK4 jacobian fp32 (512 traj x 256 knots)", cart, rng.random((131072, 5)), RD.RK4, AD, f32, a.reps)
        eval_case("large-N cartpole RK4 discrete_dynamics", cart, rng.random((1 << 22, 5)), RD.RK4, f64, 20)
        return
    if a.only == "pair":
        pair = RD.SatellitePair()
        qa, qb = rng.standard_normal((65536, 4)), rng.standard_normal((65536, 4))
        Zp = np.concatenate([qa / np.linalg.norm(qa, axis=1, keepdims=True), rng.random((65536, 3)), qb / np.linalg.norm(qb, axis=1, keepdims=True), rng.random((65536, 9))], axis=1)
        jac_case("satellite pair (n=14, m=6) RK4 jacobian ForwardAD", pair, Zp, RD.RK4, AD, f64, a.reps)
        jac_case("satellite pair (n=14, m=6) RK3 jacobian ForwardAD", pair, Zp, RD.RK3, AD, f64, a.reps)
        jac_case("satellite pair (n=14, m=6) continuous jacobian ForwardAD", pair, Zp, None, AD, f64, a.reps)
        errexp_case("satellite pair RK4 error-state expansion fused", pair, Zp, RD.RK4, f64, a.reps)
        return
    if a.only in ("multi", ""):
        if a.only == "multi":
            jac_case("C3 satellite RK4 jacobian ForwardAD (+x_next) [flagship, same box]", sat, rigid_inputs(65536, 6, rng), RD.RK4, AD, f64, a.reps)
        multi_case("16 x 4096 satellite RK4 jacobian ForwardAD", sat, lambda N: rigid_inputs(N, 6, rng), 16, 4096, RD.RK4, f64, a.reps)
        multi_case("64 x 1024 satellite RK4 jacobian ForwardAD", sat, lambda N: rigid_inputs(N, 6, rng), 64, 1024, RD.RK4, f64, a.reps)
        multi_case("32 x 101 satellite RK4 jacobian ForwardAD", sat, lambda N: rigid_inputs(N, 6, rng), 32, 101, RD.RK4, f64, a.reps)
        multi_case("1024 x 101 satellite RK4 jacobian ForwardAD (the reference's unit at scale)", sat, lambda N: rigid_inputs(N, 6, rng), 1024, 101, RD.RK4, f64, a.reps, singles=False)
        multi_case("4096 x 101 satellite RK4 jacobian ForwardAD (32 tiles per workgroup)", sat, lambda N: rigid_inputs(N, 6, rng), 4096, 101, RD.RK4, f64, a.reps, singles=False)
        multi_case("256 x 4096 satellite RK4 jacobian ForwardAD (64 tiles per workgroup)", sat, lambda N: rigid_inputs(N, 6, rng), 256, 4096, RD.RK4, f64, a.reps, singles=False)
        multi_case("1024 x 101 satellite RK4 jacobian FiniteDifference", sat, lambda N: rigid_inputs(N, 6, rng), 1024, 101, RD.RK4, f64, a.reps, singles=False, diff=FD)
        multi_case("1024 x 101 cartpole RK4 jacobian ForwardAD", cart, lambda N: rng.random((N, 5)), 1024, 101, RD.RK4, f64, a.reps, singles=False)
        multi_case("16 x 4096 cartpole RK4 jacobian ForwardAD (C2 x 16)", cart, lambda N: rng.random((N, 5)), 16, 4096, RD.RK4, f64, a.reps)
        if a.only == "multi":
            return
    if a.only == "multiprobe":     # what does the batch descriptor cost: fixed per launch or per tile?  (inline table <= 32 batches, device table beyond)
        for nbatch, per in ((1, 65536), (16, 4096), (32, 2048), (64, 1024), (1, 131072), (32, 4096), (64, 2048), (128, 1024), (1, 262144), (32, 8192), (256, 1024)):
            if nbatch == 1:
                jac_case(f"single {per}", sat, rigid_inputs(per, 6, rng), RD.RK4, AD, f64, a.reps)
            else:
                multi_case(f"{nbatch} x {per} satellite RK4 jacobian ForwardAD", sat, lambda N: rigid_inputs(N, 6, rng), nbatch, per, RD.RK4, f64, a.reps, singles=False)
        return
    if a.only == "implicit":
        Zs = rigid_inputs(65536, 6, rng)
        dmi = RD.DiscretizedDynamics(sat, RD.ImplicitMidpoint)
        Zi = RD.PackedTrajectory(13, 6, Zs, np.arange(65536) * 0.1, np.full(65536, 0.1)).to(DEV)
        Xi = torch.empty((65536, 13), dtype=torch.float64, device=DEV)
        call = bare([lambda: RD.evaluate_(dmi, Xi, Zi)])[0]        # bare C calls: the checked Python call costs more than the kernel
        us = timed(lambda i: call(), a.reps)
        emit("satellite ImplicitMidpoint discrete_dynamics (Newton)", 65536, "f64", us, 65536 * (19 * 8 + 16 + 13 * 8))
        ift = RD.ImplicitFunctionTheorem(RD.ForwardAD())
        # outputs rotated like every other Jacobian case: with ONE 136 MB output the stores land in the Infinity Cache (tools/ift_probe.py)
        jac_case("satellite ImplicitMidpoint jacobian (Newton + implicit function theorem)", sat, Zs, RD.ImplicitMidpoint, ift, f64, a.reps)
        jac_case("C3 satellite RK4 jacobian FiniteDifference", sat, Zs, RD.RK4, FD, f64, a.reps)
        jac_case("C3 satellite RK4 jacobian ForwardAD (+x_next) [flagship]", sat, Zs, RD.RK4, AD, f64, a.reps)
        return
    if a.only == "lie":
        statediff_case("quadrotor state_diff N=262144", quad, rigid_inputs(262144, 4, rng), a.reps, rng)
        statediff_case("quadrotor state_diff N=1048576", quad, rigid_inputs(1 << 20, 4, rng), max(10, a.reps // 4), rng)
        return
    if a.only == "rollout":
        rollout_case("satellite RK4 rollout fp64 4096 traj x 101 knots", sat, 4096, 101, RD.RK4, f64, 10, rng)
        rollout_case("C5 satellite RK4 rollout fp32 512 traj x 256 knots", sat, 512, 256, RD.RK4, f32, 10, rng)
        rollout_case("quadrotor RK4 rollout fp64 4096 traj x 101 knots", quad, 4096, 101, RD.RK4, f64, 10, rng)
        rollout_case("satellite RK3 rollout fp64 4096 traj x 101 knots", sat, 4096, 101, RD.RK3, f64, 10, rng)
        rollout_case("satellite RK4 rollout fp64 16384 traj x 64 knots", sat, 16384, 64, RD.RK4, f64, 10, rng)
        rollout_case("satellite RK4 rollout fp64 65536 traj x 32 knots", sat, 65536, 32, RD.RK4, f64, 10, rng)
        rollout_case("satellite RK4 rollout fp64 262144 traj x 16 knots", sat, 262144, 16, RD.RK4, f64, 5, rng)
        rollout_case("satellite RK4 rollout fp64 65536 traj x 32 knots", sat, 65536, 32, RD.RK4, f64, 10, rng, tm=True)
        rollout_case("satellite RK4 rollout fp64 262144 traj x 16 knots", sat, 262144, 16, RD.RK4, f64, 5, rng, tm=True)
        rollout_case("cartpole RK4 rollout fp64 262144 traj x 32 knots", cart, 262144, 32, RD.RK4, f64, 5, rng)
        rollout_case("cartpole RK4 rollout fp64 262144 traj x 32 knots", cart, 262144, 32, RD.RK4, f64, 5, rng, tm=True)
        rollout_jacobian_case("satellite RK4 4096 traj x 101 knots", sat, 4096, 101, RD.RK4, f64, 10, rng)
        rollout_jacobian_case("C5 satellite fp32 512 traj x 256 knots", sat, 512, 256, RD.RK4, f32, 10, rng)
        return
    if a.only == "split":
        for tm in (True, False):
            rollout_split_case("satellite RK4 fp64 32768 traj x 32 knots", sat, 32768, 32, RD.RK4, f64, 20, rng, tm=tm)
            rollout_split_case("satellite RK4 fp64 262144 traj x 16 knots", sat, 262144, 16, RD.RK4, f64, 10, rng, tm=tm)
            rollout_split_case("cartpole RK4 fp64 262144 traj x 32 knots", cart, 262144, 32, RD.RK4, f64, 10, rng, tm=tm)
            rollout_split_case("satellite RK4 fp32 65536 traj x 32 knots", sat, 65536, 32, RD.RK4, f32, 10, rng, tm=tm)
        rollout_split_case("satellite RK4 fp64 4096 traj x 101 knots (a latency-sized batch)", sat, 4096, 101, RD.RK4, f64, 10, rng, tm=True)
        return
    if a.only == "c5full":
        c5_case(10, rng, M=4096)
        rollout_case("C5 whole: cartpole RK4 rollout fp32 4096 traj x 256 knots", cart, 4096, 256, RD.RK4, f32, 10, rng, tm=True)
        rollout_case("C5 whole: satellite RK4 rollout fp32 4096 traj x 256 knots", sat, 4096, 256, RD.RK4, f32, 10, rng, tm=True)
        jac_case("C5 whole: cartpole RK4 jacobian fp32 (4096 traj x 256 knots)", cart, rng.random((1 << 20, 5)), RD.RK4, AD, f32, a.reps, with_y=False)
        jac_case("C5 whole: satellite RK4 jacobian fp32 (4096 traj x 256 knots)", sat, rigid_inputs(1 << 20, 6, rng), RD.RK4, AD, f32, a.reps, with_y=False)
        return
    if a.only == "body":
        Zs = rigid_inputs(65536, 6, rng)
        Zq = rigid_inputs(65536, 4, rng, 0.6, 1.9)
        jac_case("C3 satellite RK4 jacobian ForwardAD (+x_next) [flagship, same box]", sat, Zs, RD.RK4, AD, f64, a.reps)
        jac_case("body-frame satellite RK4 jacobian ForwardAD", RD.Satellite(bodyframe=True), Zs, RD.RK4, AD, f64, a.reps)
        jac_case("body-frame full-inertia satellite RK4 jacobian ForwardAD", RD.Satellite(mass=1.5, J=np.array([[2.0, 0.1, 0.0], [0.1, 3.0, 0.2], [0.0, 0.2, 1.0]]), bodyframe=True), Zs, RD.RK4, AD, f64, a.reps)
        jac_case("quadrotor RK4 jacobian ForwardAD [same box]", quad, Zq, RD.RK4, AD, f64, a.reps)
        jac_case("body-frame quadrotor RK4 jacobian ForwardAD", RD.Quadrotor(bodyframe=True), Zq, RD.RK4, AD, f64, a.reps)
        jac_case("body-frame satellite RK3 jacobian ForwardAD", RD.Satellite(bodyframe=True), Zs, RD.RK3, AD, f64, a.reps)
        jac_case("body-frame satellite continuous jacobian ForwardAD", RD.Satellite(bodyframe=True), Zs, None, AD, f64, a.reps)
        jac_case("body-frame satellite RK4 jacobian FiniteDifference", RD.Satellite(bodyframe=True), Zs, RD.RK4, FD, f64, a.reps)
        jac_case("body-frame satellite RK4 jacobian fp32", RD.Satellite(bodyframe=True), rigid_inputs(131072, 6, rng), RD.RK4, AD, f32, a.reps)
        return
    if a.only in ("c5", "rolljac"):
        c5_case(10, rng)
        rollout_case("C5 cartpole RK4 rollout fp32 512 traj x 256 knots", cart, 512, 256, RD.RK4, f32, 10, rng)
        rollout_case("C5 satellite RK4 rollout fp32 512 traj x 256 knots", sat, 512, 256, RD.RK4, f32, 10, rng)
        if a.only == "c5":
            return
    if a.only == "rolljac":
        rollout_jacobian_case("satellite RK4 4096 traj x 101 knots", sat, 4096, 101, RD.RK4, f64, 10, rng)
        rollout_jacobian_case("satellite RK4 16384 traj x 64 knots", sat, 16384, 64, RD.RK4, f64, 5, rng)
        rollout_jacobian_case("C5 satellite fp32 512 traj x 256 knots", sat, 512, 256, RD.RK4, f32, 10, rng)
        rollout_jacobian_case("cartpole RK4 16384 traj x 101 knots", cart, 16384, 101, RD.RK4, f64, 10, rng)
        return

    # C2: Cartpole RK4 discrete Jacobian, N = 4096
    jac_case("C2 cartpole RK4 jacobian ForwardAD", cart, rng.random((4096, 5)), RD.RK4, AD, f64, a.reps)
    # C3: Satellite RK4 dynamics + Jacobian, N = 65536
    Zs = rigid_inputs(65536, 6, rng)
    eval_case("C3 satellite RK4 discrete_dynamics", sat, Zs, RD.RK4, f64, a.reps)
    eval_case("C3 satellite continuous dynamics", sat, Zs, None, f64, a.reps)
    jac_case("C3 satellite RK4 jacobian ForwardAD (+x_next) [flagship]", sat, Zs, RD.RK4, AD, f64, a.reps)
    errexp_case("satellite RK4 error-state expansion [G2'AG1 | G2'B] fused (+x_next)", sat, rigid_inputs(65537, 6, rng), RD.RK4, f64, a.reps)
    jac_case("C3 satellite RK4 jacobian ForwardAD (J only)", sat, Zs, RD.RK4, AD, f64, a.reps, with_y=False)
    jac_case("C3 satellite RK4 jacobian UserDefined (chain rule)", sat, Zs, RD.RK4, UD, f64, a.reps)
    jac_case("C3 satellite RK4 jacobian FiniteDifference", sat, Zs, RD.RK4, FD, f64, a.reps)
    jac_case("C3 satellite continuous jacobian ForwardAD", sat, Zs, None, AD, f64, a.reps)
    jac_case("C3 satellite RK3 jacobian ForwardAD", sat, Zs, RD.RK3, AD, f64, a.reps)
    jac_case("satellite full-inertia RK4 jacobian ForwardAD", RD.Satellite(mass=1.5, J=np.array([[2.0, 0.1, 0.0], [0.1, 3.0, 0.2], [0.0, 0.2, 1.0]])), Zs, RD.RK4, AD, f64, a.reps)
    jac_case("quadrotor RK4 jacobian ForwardAD", quad, rigid_inputs(65536, 4, rng, 0.6, 1.9), RD.RK4, AD, f64, a.reps)
    pair = RD.SatellitePair()
    qa, qb = rng.standard_normal((65536, 4)), rng.standard_normal((65536, 4))
    Zp = np.concatenate([qa / np.linalg.norm(qa, axis=1, keepdims=True), rng.random((65536, 3)), qb / np.linalg.norm(qb, axis=1, keepdims=True), rng.random((65536, 9))], axis=1)
    jac_case("satellite pair (n=14, m=6) RK4 jacobian ForwardAD", pair, Zp, RD.RK4, AD, f64, a.reps)
    # C4: Quadrotor errstate Jacobian, 262144 knot points = 8 x 32768; one rank's share and the whole
    for N in (32768, 262144):
        Zq = rigid_inputs(N, 4, rng, 0.6, 1.9)
        errstate_case(f"C4 quadrotor errstate_jacobian clear=True N={N}", quad, Zq, a.reps, True)
        errstate_case(f"C4 quadrotor errstate_jacobian reference pattern-only N={N}", quad, Zq, a.reps, False)
    # C5: mixed-model Monte-Carlo, fp32: 1024 trajectories x 256 knots per GPU, half cartpole / half satellite
    nk = 512 * 256
    jac_case("C5 cartpole RK4 jacobian fp32 (512 traj x 256 knots)", cart, rng.random((nk, 5)), RD.RK4, AD, f32, a.reps)
    jac_case("C5 satellite RK4 jacobian fp32 (512 traj x 256 knots)", sat, rigid_inputs(nk, 6, rng), RD.RK4, AD, f32, a.reps)
    rollout_case("C5 satellite RK4 rollout fp32 512 traj x 256 knots", sat, 512, 256, RD.RK4, f32, 10, rng)
    rollout_case("C5 cartpole RK4 rollout fp32 512 traj x 256 knots", cart, 512, 256, RD.RK4, f32, 10, rng)
    rollout_case("satellite RK4 rollout fp64 4096 traj x 101 knots", sat, 4096, 101, RD.RK4, f64, 10, rng)
    rollout_jacobian_case("satellite RK4 4096 traj x 101 knots", sat, 4096, 101, RD.RK4, f64, 10, rng)
    rollout_jacobian_case("satellite RK4 16384 traj x 64 knots", sat, 16384, 64, RD.RK4, f64, 5, rng)
    rollout_jacobian_case("C5 satellite fp32 512 traj x 256 knots", sat, 512, 256, RD.RK4, f32, 10, rng)
    rollout_case("satellite ImplicitMidpoint rollout fp64 4096 traj x 101 knots (a Newton solve per step)", sat, 4096, 101, RD.ImplicitMidpoint, f64, 5, rng)
    # ImplicitMidpoint: Newton solve and IFT Jacobian (compute / LDS-latency bound)
    dmi = RD.DiscretizedDynamics(sat, RD.ImplicitMidpoint)
    Zi = RD.PackedTrajectory(13, 6, Zs, np.arange(65536) * 0.1, np.full(65536, 0.1)).to(DEV)
    Xi = torch.empty((65536, 13), dtype=torch.float64, device=DEV)
    call = bare([lambda: RD.evaluate_(dmi, Xi, Zi)])[0]            # bare C calls: the checked Python call costs more than the kernel
    us = timed(lambda i: call(), a.reps)
    emit("satellite ImplicitMidpoint discrete_dynamics (Newton)", 65536, "f64", us, 65536 * (19 * 8 + 16 + 13 * 8))
    ift = RD.ImplicitFunctionTheorem(RD.ForwardAD())
    jac_case("satellite ImplicitMidpoint jacobian (Newton + implicit function theorem)", sat, Zs, RD.ImplicitMidpoint, ift, f64, a.reps)
    jac_case("cartpole ImplicitMidpoint jacobian (Newton + IFT), 1M knots", cart, rng.random((1 << 20, 5)), RD.ImplicitMidpoint, ift, f64, 10)
    Zb = rigid_inputs(1 << 20, 6, rng)
    eval_case("large-N satellite RK4 discrete_dynamics", sat, Zb, RD.RK4, f64, 20)
    eval_case("large-N satellite continuous dynamics", sat, Zb, None, f64, 20)
    eval_case("large-N cartpole RK4 discrete_dynamics", cart, rng.random((1 << 22, 5)), RD.RK4, f64, 20)
    statediff_case("quadrotor state_diff N=262144", quad, rigid_inputs(262144, 4, rng, 0.6, 1.9), 30, rng)
    if a.sweep:
        for p in range(12, 23):
            N = 1 << p
            jac_case("sweep satellite RK4 jacobian ForwardAD", sat, rigid_inputs(N, 6, rng), RD.RK4, AD, f64, max(5, min(a.reps, (1 << 24) // N)))
        for p in range(12, 23, 2):
            N = 1 << p
            jac_case("sweep cartpole RK4 jacobian ForwardAD", cart, rng.random((N, 5)), RD.RK4, AD, f64, max(5, min(a.reps, (1 << 25) // N)))


if __name__ == "__main__":
    main()
