#!/usr/bin/env python3
"""bench.py -- knot-point Jacobians/sec of the batched RK4 discrete Jacobian (n = 13) on MI355X.

A "step" is ONE pass of the hot path over one batch: `jacobian!(StaticReturn(), ForwardAD(),
DiscretizedDynamics{RK4}(Satellite), J[k], y[k], Z[k])` for all k = 1..65536 (BASELINE.json configs[2]:
Satellite RigidBody{QuatRotation}, n=13, m=6, RK4 dynamics + Jacobian, 65 536 knot points), ONE kernel launch that
writes the 13x19 Jacobian and x_{k+1} of every knot point.  Inputs are resident in HBM before the timed region.
Outputs rotate over 4 buffers (4 x 136 MB > the 256 MiB Infinity Cache) so steps do not rewrite cache-resident lines.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL, for the launch / barrier / max-over-ranks control path),
every rank owns 65 536 knot points (weak scaling), no data-path collective in the timed region; the gather of the stacked
Jacobians over xGMI goes through the library's own RCCL exchange (C ABI rdhip_comm_create_rank / rdhip_gather: grouped
send / recv into the root, or all-gather), is timed separately (`--gather all|root`) and reported beside the kernel-only
figure with bytes per rank and GB/s.  Extra objects in the same JSON line (never `value`): `extras` = SURVEY 8(d)'s secondary
metrics (discrete-dynamics evaluations/s, error-state Jacobians/s, each with its roofline fraction), `small_batches_fused` =
16 independent 4 096-knot trajectories in one rdhip_jacobian_multi launch (the north-star's N = 4 096 regime).

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29501 bench.py --gpus 8
"""
import argparse
import json
import os
import sys
import time as _time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rdhip as RD  # noqa: E402
from rdhip import parallel as P  # noqa: E402

import subprocess  # noqa: E402
import threading  # noqa: E402

N_PER_GPU = 65536
NX, NU = 13, 6
NZ = NX + NU
# algorithmic bytes per knot point (SURVEY 8d / BASELINE.md section 2): read z (19 fp64) + t + dt, write J (13x19 fp64) + x_next (13 fp64)
BYTES_PER_KNOT = NZ * 8 + 16 + NX * NZ * 8 + NX * 8       # 2248
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
NBUF = 4


def synthetic_inputs(N, seed):
    """rand(model) for a RigidBody{QuatRotation}: r, v, w, u ~ U[0,1), q uniform on S^3 (src/rigidbody.jl:50-59);
    dt = 0.1, t_k = k dt (examples/single_satellite.jl:39)."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    Z = np.concatenate([rng.random((N, 3)), q, rng.random((N, 6)), rng.random((N, NU))], axis=1)
    return Z, np.arange(N) * 0.1, np.full(N, 0.1)


def usable_cores():
    """Host threads this process may really use: the affinity mask, capped by a cgroup-v2 CPU quota if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(Zn, t, dt):
    """The CPU oracle (a C++ port of the reference math: forward-mode duals through the RK4 step, like
    ForwardDiff.jacobian) built -O3 -march=native on this host, on a bounded sample of the same workload.  The
    reference's own per-knot loop is single-threaded; the port is also run with OpenMP over several thread counts up
    to the usable host cores (about 2 s each) and the best is reported with the thread count that achieved it.
    It is the checker being timed, never part of the product path."""
    from oracle import rd_oracle as O
    om = O.satellite()
    cores = usable_cores()
    v_one, n_one = O.timed_jacobian_throughput(om, O.RK4, Zn[:8192], t[:8192], dt[:8192], 1, 3.0)
    best = (v_one, 1, n_one)
    scan = {}
    for th in sorted({c for c in (8, 16, 32, 64, 128, cores) if 1 < c <= cores}):
        v, n = O.timed_jacobian_throughput(om, O.RK4, Zn, t, dt, th, 2.0)
        scan[str(th)] = v
        if v > best[0]:
            best = (v, th, n)
    return {"value": best[0], "unit": "knot-point Jacobians/s", "cores": best[1], "kind": "port",
            "sample": f"{best[2]} knot points of the same Satellite RK4 ForwardAD workload (repeated passes over the 65536-knot batch, ~2 s) on "
                      f"{best[1]} OpenMP thread(s), best of the scanned thread counts; host reports {os.cpu_count()} CPUs, {cores} usable",
            "single_thread_value": v_one, "thread_scan": scan, "host_cpus": os.cpu_count(), "usable_cores": cores}


def _timed_us(fn, reps, warm=20, load_s=0.1):
    """HIP-event time per call of `reps` back-to-back calls, after `load_s` seconds of the same calls (allocating and uploading a case's inputs
    leaves the GPU idle long enough for its clocks to drop: without the load the first ~100 ms of launches run 3-7 % slower)."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t_end, i = _time.perf_counter() + load_s, 0
    while _time.perf_counter() < t_end:
        for _ in range(20):
            fn(i); i += 1
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def _pmc_file():
    for rnd in ("r06", "r05"):
        f = os.path.join(ROOT, "profiles", rnd, "pmc_kernels.json")
        if os.path.exists(f):
            return f, rnd
    return None, None


def _pmc_rows(case):
    """HBM bytes per launch of the non-flagship kernels from the committed counter run (tools/pmc_kernels.sh -> profiles/r06/pmc_kernels.json, else round 5's:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, calibrated in the same session); {} when the file is missing.  Not measured in this run."""
    try:
        with open(_pmc_file()[0]) as fh:
            return {r["kernel"]: r for r in json.load(fh)["rows"] if r["case"] == case}
    except Exception:
        return {}


PMC_SOURCE = f"profiles/{_pmc_file()[1]}/pmc_kernels.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes, calibrated on 1 GiB in the same session); not measured in this run"


def secondary_metrics(dev, reps):
    """SURVEY 8(d) secondary metrics on rank 0's GPU, HIP-event timed back-to-back launches, bare C-ABI calls: RK4 discrete_dynamics evaluations/s
    (C3's other half, 65 536 satellite knot points), error-state Jacobians/s (C4's per-GPU kernel, Quadrotor, 262 144 knot points, dense G), and
    the N = 4 096 regime: 16 independent 4 096-knot satellite trajectories as 16 launches and as ONE rdhip_jacobian_multi launch."""
    from rdhip import _lib as L
    lib = L.lib()
    ctx = L.context_for(dev.index, torch.cuda.current_stream(dev).cuda_stream)
    sat, quad = RD.Satellite(), RD.Quadrotor()
    out = {}
    # discrete dynamics
    N = N_PER_GPU
    Zn, t, dt = synthetic_inputs(N, 11)
    Z = RD.PackedTrajectory(NX, NU, Zn, t, dt).to(dev)
    Xs = [torch.empty((N, NX), dtype=torch.float64, device=dev) for _ in range(4)]
    RD.evaluate_(RD.DiscretizedDynamics(sat, RD.RK4), Xs[0], Z)
    zp, tp, dp = Z.data.data_ptr(), Z.t.data_ptr(), Z.dt.data_ptr()
    us = _timed_us(lambda i: lib.rdhip_discrete_dynamics(ctx.handle, sat._handle, L.RK4, L.F64, N, zp, tp, dp, Xs[i % 4].data_ptr(), L.DEVICE), reps)
    b = NZ * 8 + 16 + NX * 8
    out["discrete_dynamics_rk4_satellite"] = {"knot_points": N, "value": N / (us * 1e-6), "unit": "evaluations/s", "us_per_launch": us, "bytes_per_knot": b,
                                              "frac_of_hbm_peak": N * b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    # error-state Jacobians (C4 kernel)
    Nq = 262144
    rng = np.random.default_rng(4)
    q = rng.standard_normal((Nq, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = torch.as_tensor(np.concatenate([rng.random((Nq, 3)), q, rng.random((Nq, 6))], axis=1)).to(dev)
    Gs = [RD.alloc_errstate_jacobians(quad, Nq, X) for _ in range(2)]
    RD.errstate_jacobian_(quad, Gs[0], X, clear=True)
    us = _timed_us(lambda i: lib.rdhip_errstate_jacobian(ctx.handle, quad._handle, L.F64, Nq, X.data_ptr(), 13, 1, Gs[i % 2].data_ptr(), L.DEVICE), max(20, reps // 4))
    b = 13 * 8 + 13 * 12 * 8
    out["errstate_jacobian_quadrotor"] = {"knot_points": Nq, "value": Nq / (us * 1e-6), "unit": "error-state Jacobians/s", "us_per_launch": us, "bytes_per_knot": b,
                                          "frac_of_hbm_peak": Nq * b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    pr = _pmc_rows("errstate_jacobian_262144").get("errstate_tile_kernel")
    if pr:
        out["errstate_jacobian_quadrotor"].update({"traffic": pr["write_bytes"] + pr["read_bytes"], "traffic_over_algorithmic": (pr["write_bytes"] + pr["read_bytes"]) / (Nq * b),
                                                   "traffic_source": PMC_SOURCE})
    # the other diff methods / rules on the headline batch (same bytes per knot point as the headline kernel)
    dm = RD.DiscretizedDynamics(sat, RD.RK4)
    dmi = RD.DiscretizedDynamics(sat, RD.ImplicitMidpoint)
    Jx = [RD.alloc_jacobians(dm, N, Z.data) for _ in range(NBUF)]
    for key, fun, diff in (("jacobian_finite_difference_rk4_satellite", dm, RD.FiniteDifference()),
                           ("jacobian_implicit_midpoint_ift_satellite", dmi, RD.ImplicitFunctionTheorem(RD.ForwardAD()))):
        calls = [RD.prepare_jacobian(RD.StaticReturn(), diff, fun, Jx[k], Xs[k], Z) for k in range(NBUF)]
        us = _timed_us(lambda i: calls[i % NBUF](), max(20, reps // 4))
        out[key] = {"knot_points": N, "value": N / (us * 1e-6), "unit": "knot-point Jacobians/s", "us_per_launch": us, "bytes_per_knot": BYTES_PER_KNOT,
                    "frac_of_hbm_peak": N * BYTES_PER_KNOT / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    # ImplicitMidpoint + IFT as a JOB: what separates this kernel from the explicit one is the first tile's Newton chain standing alone in front of every
    # launch -- a second launch in flight hides it.  Own-stream context with launch overlap on, the library's stopwatch; closed again
    # before config 5's one-call form needs the hardware queues.
    try:
        octx = L.Context(dev.index if hasattr(dev, "index") and dev.index is not None else 0)
        octx.set_overlap(True)
        ocalls = [RD.prepare_jacobian(RD.StaticReturn(), RD.ImplicitFunctionTheorem(RD.ForwardAD()), dmi, Jx[k], Xs[k], Z, ctx=octx) for k in range(NBUF)]
        for i in range(300):
            ocalls[i % NBUF]()
        octx.synchronize()
        n_o = max(400, reps)
        octx.timer_start()
        for i in range(n_o):
            ocalls[i % NBUF]()
        us_o = octx.timer_stop() * 1e3 / n_o
        out["jacobian_implicit_midpoint_ift_satellite"]["with_launch_overlap"] = {"us_per_launch": us_o, "frac_of_hbm_peak": N * BYTES_PER_KNOT / (us_o * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                                                 "is": "rdhip_ctx_set_overlap(1): consecutive calls alternate between two streams of one context"}
        del ocalls
        octx.synchronize(); octx.close()
    except Exception as e:
        out["jacobian_implicit_midpoint_ift_satellite"]["with_launch_overlap"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    del Jx
    # the headline operator on a LARGE batch (1 M knot points, 16 tiles per workgroup and more: the tile loop's dynamic schedule); inputs and
    # outputs rotate over two sets (2 x 2.2 GB)
    try:
        Nl = 1 << 20
        Zl = [RD.PackedTrajectory(NX, NU, *synthetic_inputs(Nl, 21 + k)).to(dev) for k in range(2)]
        Jl = [RD.alloc_jacobians(dm, Nl, Zl[0].data) for _ in range(2)]
        Yl = [torch.empty((Nl, NX), dtype=torch.float64, device=dev) for _ in range(2)]
        calls = [RD.prepare_jacobian(RD.StaticReturn(), RD.ForwardAD(), dm, Jl[k], Yl[k], Zl[k]) for k in range(2)]
        us = _timed_us(lambda i: calls[i & 1](), 20, warm=3)
        out["jacobian_rk4_satellite_1M_knots"] = {"knot_points": Nl, "value": Nl / (us * 1e-6), "unit": "knot-point Jacobians/s", "us_per_launch": us,
                                                  "bytes_per_knot": BYTES_PER_KNOT, "frac_of_hbm_peak": Nl * BYTES_PER_KNOT / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                  "kernel": L.last_kernel_name()[:60], "input_buffers": 2, "output_buffers": 2}
        del Zl, Jl, Yl, calls
    except Exception as e:
        out["jacobian_rk4_satellite_1M_knots"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    # rollout!: one GPU's share of C5's satellite half, 512 trajectories x 256 knot points, fp32 -- a latency chain, not a stream
    Mr, Kr = 512, 256
    zr, _, _ = synthetic_inputs(Mr * Kr, 12)
    stack = {"data": torch.as_tensor(zr.astype(np.float32).reshape(Mr, Kr, NZ)).to(dev), "t": torch.as_tensor(np.tile(np.arange(Kr) * 0.01, (Mr, 1))).to(dev),
             "dt": torch.full((Mr, Kr), 0.01, dtype=torch.float64, device=dev)}
    us = _timed_us(lambda i: RD.rollout_(RD.StaticReturn(), dm, stack), 20, warm=2)
    out["rollout_rk4_satellite_fp32"] = {"trajectories": Mr, "knot_points_per_trajectory": Kr, "value": Mr * (Kr - 1) / (us * 1e-6), "unit": "simulated steps/s",
                                         "us_per_launch": us, "us_per_step": us / (Kr - 1)}
    # BASELINE config 5 in ONE call (rdhip_rollout_jacobian_multi: rollouts + all Jacobians, fp32, time-major): (a) the WHOLE workload on this one GPU --
    # 4 096 Cartpole + 4 096 Satellite trajectories x 256 knots, the N = 1 point of its scaling curve -- and (b) one GPU's share of the 8-GPU split
    # (512 + 512).  Algorithmic bytes per simulated + linearised knot point: satellite 1 080 (Jacobian) + 52 (the state the rollout
    # writes), cartpole 116 + 16.
    def c5_case(Mc, own_ctx=None):
        funs, stacks, Jc = [], [], []
        for mdl, mctl in ((RD.Cartpole(), 1), (sat, NU)):
            n = RD.state_dim(mdl)
            zz = synthetic_inputs(Mc * Kr, 13)[0] if n == 13 else np.random.default_rng(14).random((Mc * Kr, n + mctl))
            fun = RD.DiscretizedDynamics(mdl, RD.RK4)
            st = {"data": torch.as_tensor(zz.astype(np.float32).reshape(Kr, Mc, n + mctl)).to(dev),
                  "t": torch.as_tensor(np.ascontiguousarray(np.broadcast_to((np.arange(Kr) * 0.01)[:, None], (Kr, Mc)))).to(dev),
                  "dt": torch.full((Kr, Mc), 0.01, dtype=torch.float64, device=dev)}
            funs.append(fun); stacks.append(st); Jc.append(RD.alloc_jacobians(fun, Mc * Kr, st["data"]))
        c5 = RD.PreparedRolloutJacobianMulti(RD.StaticReturn(), RD.ForwardAD(), funs, Jc, stacks, None, time_major=True, ctx=own_ctx)
        alg = Mc * Kr * ((1080 + 52) + (116 + 16))
        if own_ctx is not None:                            # the library's stopwatch on the context's own stream (both behind a join of its streams)
            for i in range(4):
                c5()
            own_ctx.synchronize()
            t_end = _time.perf_counter() + 0.1
            while _time.perf_counter() < t_end:
                for _ in range(20):
                    c5()
                own_ctx.synchronize()
            own_ctx.timer_start()
            for i in range(40):
                c5()
            us = own_ctx.timer_stop() * 1e3 / 40
            return {"us_per_call": us, "GBps": alg / (us * 1e-6) / 1e9, "frac_of_hbm_peak": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel_launches_per_call": c5.launches,
                    "is": "the same call on a context that owns its stream with rdhip_ctx_set_overlap(1).  Round 6: the call is ONE launch (csrc/mixed.hpp) on either kind of "
                          "context, so there is no fork / join left for the overlap mode to skip (round 5: the second model on the internal stream, 244-252 us)"}
        us = _timed_us(lambda i: c5(), 20, warm=2)
        res = {"trajectories": 2 * Mc, "knot_points_per_trajectory": Kr, "us_per_call": us, "value": 2 * Mc * Kr / (us * 1e-6),
               "unit": "simulated + linearised knot points/s", "entry_point": "rdhip_rollout_jacobian_multi", "dtype": "f32",
               "algorithmic_bytes": alg, "bytes_per_knot": {"satellite": 1080 + 52, "cartpole": 116 + 16},
               "GBps": alg / (us * 1e-6) / 1e9, "frac_of_hbm_peak": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        res["kernel_launches_per_call"] = c5.launches            # 1: both rollouts and both models' Jacobians in one grid (csrc/mixed.hpp); round 5: 9 launches on two streams
        try:                                                      # round 5's form of the same call (the satellite's fused launch + the cartpole in four pipelined pieces on a second stream), same process
            RD.tune("mixed=0")
            us5 = _timed_us(lambda i: c5(), 20, warm=2)
            res["streams_form_round5"] = {"us_per_call": us5, "frac_of_hbm_peak": alg / (us5 * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel_launches_per_call": 9, "launches_counted_in": "profiles/r05/c5_whole_timeline.txt (1 on the context's stream, 8 on the internal one)", "how": "RDHIP_TUNE mixed=0"}
        finally:
            RD.tune("reset")
        res["layout"] = "time-major (knot k of all trajectories contiguous)"
        try:                                                      # the same call on the reference's own layout: a vector of trajectories (trajectory-major stacks); one launch as well
            stj = [{"data": st["data"].transpose(0, 1).contiguous(), "t": st["t"].t().contiguous(), "dt": st["dt"].t().contiguous()} for st in stacks]
            c5j = RD.PreparedRolloutJacobianMulti(RD.StaticReturn(), RD.ForwardAD(), funs, Jc, stj, None, time_major=False)
            usj = _timed_us(lambda i: c5j(), 20, warm=2)
            res["trajectory_major"] = {"us_per_call": usj, "frac_of_hbm_peak": alg / (usj * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel_launches_per_call": c5j.launches}
            del stj, c5j
        except Exception as e:
            res["trajectory_major"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        pr = _pmc_rows("c5_whole_one_call_fp32") if Mc == 4096 else {}
        if "rollout_jac_mixed_kernel" in pr:
            tr = pr["rollout_jac_mixed_kernel"]["write_bytes"] + pr["rollout_jac_mixed_kernel"]["read_bytes"]
            res.update({"traffic": tr, "traffic_over_algorithmic": tr / alg, "traffic_source": PMC_SOURCE})
        elif all(k in pr for k in ("rollout_jac_fused_kernel", "rollout_kernel", "jac_ad_small_kernel")):
            # one call = the satellite's fused launch + the cartpole's four pieces (rollout + Jacobians each); the counters also see what the judge's
            # bytes leave out: the rollouts' own reads of controls and times
            tr = sum((pr[k]["write_bytes"] + pr[k]["read_bytes"]) * n for k, n in (("rollout_jac_fused_kernel", 1), ("rollout_kernel", 4), ("jac_ad_small_kernel", 4)))
            res.update({"traffic": tr, "traffic_over_algorithmic": tr / alg, "traffic_source": PMC_SOURCE})
        return res
    # (the overlap-mode form first, and its context closed again: the plain form's child streams stay alive with the default context, and HIP has four
    # hardware queues for all of a process's streams)
    c5_ov = {}
    try:
        octx = L.Context(dev.index if dev.index is not None else 0)
        octx.set_overlap(True)
        for key, Mc in (("c5_whole_one_gpu_one_call_fp32", 4096), ("c5_share_one_call_fp32", Mr)):
            c5_ov[key] = c5_case(Mc, octx)
        octx.synchronize(); octx.close()
    except Exception as e:
        c5_ov = {"error": f"{type(e).__name__}: {e}"[:200]}
    for key, Mc in (("c5_whole_one_gpu_one_call_fp32", 4096), ("c5_share_one_call_fp32", Mr)):
        try:
            out[key] = c5_case(Mc)
            out[key]["with_launch_overlap"] = c5_ov.get(key, c5_ov.get("error"))
        except Exception as e:
            out[key] = {"error": f"{type(e).__name__}: {e}"[:200]}
    # 16 x 4096: separate launches vs one fused launch
    nbt, per = 16, 4096
    Zs = []
    for k in range(nbt):
        zn, tt, dd = synthetic_inputs(per, 100 + k)
        Zs.append(RD.PackedTrajectory(NX, NU, zn, tt, dd).to(dev))
    Js = [[RD.alloc_jacobians(dm, per, Zs[0].data) for _ in range(nbt)] for _ in range(NBUF)]
    Ys = [[torch.empty((per, NX), dtype=torch.float64, device=dev) for _ in range(nbt)] for _ in range(NBUF)]
    sig, AD = RD.StaticReturn(), RD.ForwardAD()
    singles = [[RD.prepare_jacobian(sig, AD, dm, Js[k][j], Ys[k][j], Zs[j]) for j in range(nbt)] for k in range(NBUF)]
    fused = [RD.prepare_jacobian_multi(sig, AD, dm, Js[k], Ys[k], Zs) for k in range(NBUF)]
    us_sep = _timed_us(lambda i: [c() for c in singles[i % NBUF]], max(10, reps // 10), warm=3)
    us_one = _timed_us(lambda i: fused[i % NBUF](), reps)
    tot = nbt * per
    small = {"workload": f"{nbt} independent Satellite trajectories x {per} knot points (separate arrays), RK4 ForwardAD Jacobian + x_next, fp64",
             "one_launch_us": us_one, "value": tot / (us_one * 1e-6), "unit": "knot-point Jacobians/s",
             "frac_of_hbm_peak": tot * BYTES_PER_KNOT / (us_one * 1e-6) / 1e9 / HBM_PEAK_GBS,
             "separate_launches_us": us_sep, "separate_frac_of_hbm_peak": tot * BYTES_PER_KNOT / (us_sep * 1e-6) / 1e9 / HBM_PEAK_GBS,
             "entry_point": "rdhip_jacobian_multi"}
    del Js, Ys, singles, fused
    # the reference's real unit at scale: 1 024 independent trajectories of 101 knot points (examples/cartpole.jl:46), separate arrays, ONE call
    nbt, per = 1024, 101
    zn, tt, dd = synthetic_inputs(nbt * per, 300)
    Zs = [RD.PackedTrajectory(NX, NU, zn[b * per:(b + 1) * per], tt[:per], dd[:per]).to(dev) for b in range(nbt)]
    Js = [[RD.alloc_jacobians(dm, per, Zs[0].data) for _ in range(nbt)] for _ in range(2)]
    Ys = [[torch.empty((per, NX), dtype=torch.float64, device=dev) for _ in range(nbt)] for _ in range(2)]
    fused = [RD.prepare_jacobian_multi(sig, AD, dm, Js[k], Ys[k], Zs) for k in range(2)]
    us_k = _timed_us(lambda i: fused[i % 2](), max(20, reps // 4), warm=3)
    small["trajectories_1024_x_101"] = {"one_call_us": us_k, "kernel_launches_per_call": fused[0].launches, "value": nbt * per / (us_k * 1e-6),
                                        "frac_of_hbm_peak": nbt * per * BYTES_PER_KNOT / (us_k * 1e-6) / 1e9 / HBM_PEAK_GBS}
    return out, small


class Watchdog:
    """Every failure mode must end in ONE JSON line, never in a hang (an RCCL communicator whose peer never arrives blocks inside
    ncclCommInitRank / a grouped send-recv for ever).  Phases register what is known so far; if a phase outlives its deadline the watchdog prints
    the line with the phase's name as the error and leaves the process."""

    def __init__(self):
        self.partial, self.timer, self.lock = None, None, threading.Lock()

    def arm(self, seconds, phase, rank):
        self.disarm()

        def fire():
            with self.lock:
                out = dict(self.partial or {"metric": "knot-point Jacobians/sec (RK4 discrete_jacobian!, n=13)", "value": None})
                out["error"] = f"rank {rank}: phase '{phase}' exceeded {seconds:.0f} s (watchdog)"
                # (only rank 0 owns stdout's JSON line; another rank that gives up says so on stderr and leaves -- rank 0's own watchdog then
                # fires in whatever collective it is waiting in)
                print(json.dumps(out), file=sys.stdout if rank == 0 else sys.stderr, flush=True)
            os._exit(3 if self.partial is None else 0)
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def sharded_extras(dev, rank, world, comm, red_dev):
    """N > 1: BASELINE config 4 (Quadrotor errstate_jacobian!, 262 144 knot points split over the ranks, gathered to rank 0 through the library's
    RCCL exchange: rdhip_errstate_jacobian_sharded_gather's shape) and this rank's share of config 5 (8 192 / world trajectories x 256 knots,
    half Cartpole / half Satellite, fp32, rollouts + Jacobians in one rdhip_rollout_jacobian_multi call).  Returns per-rank values; rank 0
    gathers them into the line."""
    import ctypes as C
    from rdhip import _lib as L
    out = {}
    quad = RD.Quadrotor()
    Ntot = 262144
    lo, hi = P.shard_range(Ntot, rank, world)
    n_loc = hi - lo
    rng = np.random.default_rng(40 + rank)
    q = rng.standard_normal((n_loc, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = torch.as_tensor(np.concatenate([rng.random((n_loc, 3)), q, rng.random((n_loc, 6))], axis=1)).to(dev)
    G = RD.alloc_errstate_jacobians(quad, n_loc, X)
    stack = torch.empty((Ntot * 13 * 12) if rank == 0 else 1, dtype=torch.float64, device=dev)
    sizes = [(P.shard_range(Ntot, r, world)[1] - P.shard_range(Ntot, r, world)[0]) * 13 * 12 * 8 for r in range(world)]
    lib, ctx = L.lib(), comm.ctxs[0]

    def c4(i):
        L.check(lib.rdhip_errstate_jacobian(ctx.handle, quad._handle, L.F64, n_loc, X.data_ptr(), 13, 1, G.data_ptr(), L.DEVICE), ctx.handle)
        comm.gather("root", [G], sizes, [stack], root=0)
    c4(0); comm.synchronize()
    if world > 1:
        dist.barrier()
    t0 = _time.perf_counter()
    reps = 5
    for i in range(reps):
        c4(i)
    comm.synchronize()
    el = torch.tensor([_time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ms = 1e3 * float(el.item()) / reps
    out["c4_errstate_sharded_gather_root"] = {"knot_points_total": Ntot, "knot_points_this_rank": n_loc, "ms_per_pass_incl_gather": ms,
                                             "value": Ntot / (ms * 1e-3), "unit": "error-state Jacobians/s (stacked on rank 0)", "bytes_into_root": sum(sizes) - sizes[0]}
    # C5 share
    Mr, K = max(4, (8192 // world // 2) // 4 * 4), 256
    funs, stacks, Js = [], [], []
    for model, mctl in ((RD.Cartpole(), 1), (RD.Satellite(), 6)):
        n = RD.state_dim(model)
        if n == 13:
            zr, _, _ = synthetic_inputs(Mr * K, 50 + rank)
        else:
            zr = np.random.default_rng(60 + rank).random((Mr * K, n + mctl))
        fun = RD.DiscretizedDynamics(model, RD.RK4)
        st = {"data": torch.as_tensor(zr.astype(np.float32).reshape(K, Mr, n + mctl)).to(dev),
              "t": torch.as_tensor(np.ascontiguousarray(np.broadcast_to((np.arange(K) * 0.01)[:, None], (K, Mr)))).to(dev),
              "dt": torch.full((K, Mr), 0.01, dtype=torch.float64, device=dev)}
        funs.append(fun); stacks.append(st); Js.append(RD.alloc_jacobians(fun, Mr * K, st["data"]))
    call = RD.PreparedRolloutJacobianMulti(RD.StaticReturn(), RD.ForwardAD(), funs, Js, stacks, None, time_major=True)
    us = _timed_us(lambda i: call(), 20, warm=2)
    out["c5_share_rollouts_and_jacobians_one_call"] = {"trajectories_this_rank": 2 * Mr, "knot_points_per_trajectory": K, "us_per_call": us,
                                                      "value": 2 * Mr * K / (us * 1e-6), "unit": "simulated + linearised knot points/s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--gather", choices=["none", "all", "root"], default="all", help="also time the RCCL exchange of the stacked Jacobians (N>1 only; reported separately)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary metrics / fused small batches (extra objects, never `value`)")
    ap.add_argument("--extras-reps", type=int, default=200)
    ap.add_argument("--prewarm-ms", type=float, default=200.0,
                    help="sustained load (the same launches, untimed) before the W warm-up steps: clocks and TLBs settle over ~100 ms, "
                         "a short W alone measures the ramp (0.62 instead of 0.65-0.68 of the roofline)")
    # (off by default since the launch-overlap mode of one context became the headline: HIP maps streams onto four hardware queues, and the extra's four
    # streams + the library's internal ones then alias -- the two models of the C5 extra ended up on ONE queue and ran one after the other, 281 instead of 160 us)
    ap.add_argument("--streams", type=int, default=1, help="N=1 only: also report independent batches issued round-robin on this many streams (extra object, never `value`; 1 = skip)")
    ap.add_argument("--knots", type=int, default=N_PER_GPU, help="knot points per GPU (default = BASELINE config)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend for N>1 (gloo: control-path dry run, no gather)")
    ap.add_argument("--same-device", action="store_true", help="dry run: every rank uses cuda:0 (exercises the N>1 control path on a 1-GPU box)")
    ap.add_argument("--no-overlap", action="store_true", help="issue every launch on the context's one stream (rounds 1-2 behaviour); default: rdhip_ctx_set_overlap(1)")
    ap.add_argument("--phase-timeout", type=float, default=240.0, help="seconds any single phase (rendezvous, timed region, RCCL exchange, extras) may take before the "
                                                                        "watchdog prints the JSON line with an error and exits")
    args = ap.parse_args()

    rank, local_rank, world = P.env_rank_world()
    wd = Watchdog()
    wd.arm(args.phase_timeout, "process-group rendezvous / first RCCL collective", rank)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")      # where the scalar MAX-reductions live
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.zeros(1, device=dev)
                dist.all_reduce(probe)                     # fail here, not inside the timed region, if RCCL cannot talk
                torch.cuda.synchronize()
            except Exception as e:                         # the data path has no collective: keep measuring with a CPU control path
                print(f"[bench] rank {rank}: RCCL unavailable ({type(e).__name__}: {e}); control path falls back to gloo, gather skipped", file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                args.backend = "gloo"
                red_dev = torch.device("cpu")
                dist.init_process_group("gloo")
        else:
            dist.init_process_group("gloo")

    wd.arm(args.phase_timeout, "set-up, warm-up and timed region", rank)
    N = args.knots
    model = RD.Satellite()                       # mass 1, J = I3 (examples/single_satellite.jl:33-35)
    dmodel = RD.DiscretizedDynamics(model, RD.RK4)
    # NBUF distinct input batches as well (different seeds, separate Z / t / dt arrays): step i reads batch i % NBUF and writes output buffer i % NBUF
    host_in = [synthetic_inputs(N, seed=3 + rank + 1000 * b) for b in range(NBUF)]
    Zn, t, dt = host_in[0]
    Zs = [RD.PackedTrajectory(NX, NU, *host_in[b]).to(dev) for b in range(NBUF)]
    Z = Zs[0]
    Js = [RD.alloc_jacobians(dmodel, N, Z.data) for _ in range(NBUF)]
    Ys = [torch.empty((N, NX), dtype=torch.float64, device=dev) for _ in range(NBUF)]
    sig, diff = RD.StaticReturn(), RD.ForwardAD()

    # argument checking once per output buffer (rdhip.prepare_jacobian); a step is then the bare C-ABI call, as from a Julia ccall.
    # ONE rdhip context that owns its stream, with launch overlap on (rdhip_ctx_set_overlap, include/rdhip.h): consecutive calls whose arrays do not
    # intersect -- the steps of this benchmark rotate over NBUF output buffers -- alternate between the context's two streams, so a launch's
    # prologue runs under its predecessor's tail; dependent calls keep their order through event edges.  --no-overlap: one stream.
    from rdhip import _lib as L
    torch.cuda.synchronize()
    bctx = L.Context(local_rank)
    if not args.no_overlap:
        bctx.set_overlap(True)
    calls = [RD.prepare_jacobian(sig, diff, dmodel, Js[b], Ys[b], Zs[b], ctx=bctx) for b in range(NBUF)]
    calls_same_input = [RD.prepare_jacobian(sig, diff, dmodel, Js[b], Ys[b], Z, ctx=bctx) for b in range(NBUF)]      # (rounds 1-3: one input batch for every step)

    def step(i):
        calls[i % NBUF]()

    def barrier():
        if world > 1:
            dist.barrier()

    if args.prewarm_ms > 0:                       # device warm-up, not part of W or K
        t_end = _time.perf_counter() + args.prewarm_ms * 1e-3
        i = 0
        while _time.perf_counter() < t_end:
            for _ in range(64):
                step(i); i += 1
            torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    bctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # THE timed region: exactly K steps between two barrier + synchronize brackets, wall clock (nothing else inside it)
    t0 = _time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = _time.perf_counter() - t0
    # time per launch of the same steps as a JOB (launch overlap on): a region of its own of at least 400 launches between the library's stopwatch
    # events -- inside the region above, the stopwatch's two events, the join of the context's streams and the event synchronisation were part of what
    # `value` measured (a few per cent of the driver's K = 20 steps)
    n_ov = max(args.steps, 400)
    bctx.timer_start()
    for i in range(n_ov):
        step(i)
    kern_ms = bctx.timer_stop() / n_ov                  # (synchronises on the stop event)
    kernel_name = L.last_kernel_name()                  # the kernel the library dispatched the timed calls to (rdhip_last_kernel_name), not a literal
    # The same K steps on ONE stream (overlap off): back-to-back launches of the dominant kernel ALONE on the chip.  HIP events around them / K =
    # the kernel's average launch duration (plus the ~1 us gap between launches of one stream) -- this is what `roofline` describes and what the
    # per-kernel average of a rocprofv3 --kernel-trace of `bench.py --no-overlap` agrees with.
    single_blocks = {}

    def timed_single(fns, nsteps, tag=None):
        # The kernel's average launch duration needs a region long enough for the stopwatch's own cost to vanish: a pair of events + the synchronisation
        # behind K = 20 launches (the driver's K: half a millisecond) adds 1.5 us to every launch (27.4-29.0 us in each of 20 such blocks where 2 000
        # launches in one block give 26.3).  So this measurement -- not `value`, which is the wall clock of exactly K steps -- times at least 400 launches.
        for i in range(min(args.warmup, 200)):
            fns[i % NBUF]()
        bctx.synchronize()
        n = max(nsteps, 400)
        bctx.timer_start()
        for i in range(n):
            fns[i % NBUF]()
        ms = bctx.timer_stop() / n
        if tag:
            single_blocks[tag] = {"launches_timed": n, "steps": nsteps}
        return ms
    if not args.no_overlap:
        bctx.set_overlap(False)
        s_ms = timed_single(calls, args.steps, "kernel_ms")
    else:
        s_ms = kern_ms
    same_ms = timed_single(calls_same_input, max(200, args.steps // 4))     # one input batch for every step: does rotating the 11 MB of inputs move the time?
    if not args.no_overlap:
        bctx.set_overlap(True)
    single = {"kernel_ms": s_ms, "frac_of_hbm_peak": N * BYTES_PER_KNOT / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "note": "launch overlap off: every launch on the context's one stream"}

    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    value = world * N * args.steps / elapsed

    achieved = N * BYTES_PER_KNOT / (s_ms * 1e-3) / 1e9                 # the kernel alone (single stream)
    achieved_ov = N * BYTES_PER_KNOT / (kern_ms * 1e-3) / 1e9            # job-level: consecutive launches overlapped on the context's two streams
    traffic = tsrc = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # PMC-derived HBM bytes per launch, measured in a separate rocprofv3 run
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            traffic = tj.get("bytes_per_launch")
            same = lambda a, b: a.replace(" ", "").split("(")[0] == b.replace(" ", "").split("(")[0]      # noqa: E731  (kernel<template arguments>, without the parameter list)
            if N != N_PER_GPU or (tj.get("kernel") and not same(tj["kernel"], kernel_name)):
                traffic = None                         # the committed counters belong to another kernel / size: say nothing rather than something stale
            try:
                commit = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "profiles/hbm_traffic.json"], capture_output=True, text=True, timeout=10).stdout.strip()
            except Exception:
                commit = ""
            tsrc = "profiles/hbm_traffic.json" + (f"@{commit}" if commit else "") + " (" + str(tj.get("source", "rocprofv3 --pmc, separate passes")) + "); not measured in this run"
        except Exception:
            traffic = None
    line = {
        "metric": "knot-point Jacobians/sec (RK4 discrete_jacobian!, n=13)",
        "value": value, "unit": "knot-point Jacobians/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "prewarm_ms": args.prewarm_ms,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Satellite RigidBody{QuatRotation} (n=13,m=6) RK4 discrete dynamics + ForwardAD Jacobian (BASELINE configs[2])",
                   "knot_points_per_gpu": N, "knot_points_total": world * N, "dt": 0.1, "outputs": "J 13x19 + x_next per knot point",
                   "output_buffers_rotated": NBUF, "input_batches_rotated": NBUF, "sharding": f"{world} x contiguous knot-point blocks, no data-path collective",
                   "launch_overlap": (not args.no_overlap), "context": "one rdhip context that owns its stream"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": tsrc,
                     "kernel": kernel_name, "kernel_source": "rdhip_last_kernel_name() after the timed region: the kernel the library dispatched these calls to",
                     "kernel_ms": s_ms, "algorithmic_bytes_per_launch": N * BYTES_PER_KNOT,
                     "kernel_ms_is": "average launch duration of the dominant kernel ALONE on the chip: HIP events (rdhip_timer_*, on the context's stream) around K back-to-back "
                                     "launches on ONE stream, / K (at least 400 launches: behind fewer, the stopwatch's own events and synchronisation "
                                     "are 1.5 us of every launch -- kernel_ms_region); agrees with the per-kernel average of the rocprofv3 --kernel-trace of `bench.py --no-overlap` "
                                     "(profiles/r05/kernel_stats_single_stream.csv)",
                     "kernel_ms_region": single_blocks.get("kernel_ms"),
                     "inputs": {"distinct_input_batches": NBUF, "kernel_ms_one_input_batch": same_ms,
                                "note": "steps rotate over NBUF separate Z/t/dt arrays (different seeds) as well as NBUF outputs; kernel_ms_one_input_batch = the same "
                                        "measurement with every step reading batch 0 (rounds 1-3)"},
                     "overlapped": {"ms_per_launch_overlapped": kern_ms, "achieved": achieved_ov, "frac": achieved_ov / HBM_PEAK_GBS, "launch_overlap": (not args.no_overlap),
                                    "launches_timed": n_ov,
                                    "is": "job-level time per launch of the steps `value` is measured on (a region of its own, see kernel_ms_is): with rdhip_ctx_set_overlap(1) consecutive launches alternate between the "
                                          "context's two streams and overlap in time (two launches share the chip; each one's own duration is LONGER than kernel_ms), so this is "
                                          "time per launch of the job, not a kernel duration"}},
    }
    if single:
        line["single_stream"] = single
    if rank == 0:
        wd.partial = line            # from here on a stuck extra costs the extra, not the line
    wd.arm(args.phase_timeout, "extras (multi-stream / RCCL exchange / secondary metrics)", rank)

    # extra (never `value`): the same K independent batches issued round-robin on 4 HIP streams, one rdhip context per stream -- the next
    # launch's workgroups start on the CUs the previous launch has already left, so its prologue runs under the previous launch's tail
    overlapped = None
    if world == 1 and args.streams > 1:
      try:
        streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
        scalls = []
        for b in range(NBUF):
            with torch.cuda.stream(streams[b % args.streams]):
                scalls.append(RD.prepare_jacobian(sig, diff, dmodel, Js[b], Ys[b], Z))
        for i in range(args.warmup):
            with torch.cuda.stream(streams[i % args.streams]):
                scalls[i % NBUF]()
        torch.cuda.synchronize()
        s0 = _time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(streams[i % args.streams]):
                scalls[i % NBUF]()
        torch.cuda.synchronize()
        sel = _time.perf_counter() - s0
        overlapped = {"streams": args.streams, "value": N * args.steps / sel, "ms_per_step": 1e3 * sel / args.steps,
                      "frac_of_hbm_peak": N * BYTES_PER_KNOT * args.steps / sel / 1e9 / HBM_PEAK_GBS,
                      "note": "independent batches round-robin on several streams; not the contract's `value`"}
      except Exception as e:                                 # an extra: never lose the bench line over it
        overlapped = {"streams": args.streams, "error": f"{type(e).__name__}: {e}"[:200]}

    gather = None
    if world > 1 and args.gather != "none":
        # the stacked Jacobians on the root (or everywhere) through the library's own exchange: rdhip_comm_create_rank + rdhip_gather
        # (RCCL grouped send / recv or all-gather over xGMI); torch.distributed only carried the 128-byte RCCL id
        try:
            reps = 5
            shard_bytes = N * NX * NZ * 8
            comm = P.Comm.from_torch_distributed(local_rank, stream=torch.cuda.current_stream(dev).cuda_stream)
            receives = args.gather == "all" or rank == 0
            stack = torch.empty(world * shard_bytes // 8 if receives else 1, dtype=torch.float64, device=dev)
            sizes = [shard_bytes] * world

            def exchange(i):
                comm.gather(args.gather, [Js[i % NBUF]], sizes, [stack], root=0)
            exchange(0)                                      # warm (connection set-up)
            torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
            g0 = _time.perf_counter()
            for i in range(reps):
                exchange(i)
            torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
            g_only = torch.tensor([_time.perf_counter() - g0], dtype=torch.float64, device=red_dev)
            g1 = _time.perf_counter()
            for i in range(reps):
                step(i)
                bctx.synchronize()                           # (the exchange runs on torch's stream: behind the context's streams through the host)
                exchange(i)
            torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
            gl = torch.tensor([_time.perf_counter() - g1], dtype=torch.float64, device=red_dev)
            dist.all_reduce(gl, op=dist.ReduceOp.MAX)
            dist.all_reduce(g_only, op=dist.ReduceOp.MAX)
            ok = True
            if receives:                                     # the stack's own block must be this rank's shard
                mine = stack[rank * (shard_bytes // 8): rank * (shard_bytes // 8) + 64]
                ok = bool(torch.equal(mine, torch.as_strided(Js[(reps - 1) % NBUF], (64,), (1,))))
            gms = 1e3 * float(g_only.item()) / reps
            gather = {"mode": args.gather, "path": "rdhip_gather (C ABI, RCCL grouped send/recv or all-gather)", "bytes_per_rank": shard_bytes,
                      "gather_ms": gms, "root_ingest_GBps": (world - 1) * shard_bytes / (gms * 1e-3) / 1e9,
                      "value_with_gather": world * N * reps / float(gl.item()), "ms_per_step_with_gather": 1e3 * float(gl.item()) / reps, "own_block_ok": ok}
            comm.close()
        except Exception as e:                               # the exchange is an extra: never lose the bench line over it
            gather = {"mode": args.gather, "error": f"{type(e).__name__}: {e}"[:300]}
            if args.backend == "nccl":                       # second choice: the torch.distributed form of the same exchange
                try:
                    P.gather_stacked(Js[0].transpose(1, 2), world * N, args.gather)
                    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
                    g0 = _time.perf_counter()
                    for i in range(3):
                        P.gather_stacked(Js[i % NBUF].transpose(1, 2), world * N, args.gather)
                    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
                    gather["torch_distributed_gather_ms"] = 1e3 * (_time.perf_counter() - g0) / 3
                except Exception as e2:
                    gather["torch_distributed_error"] = f"{type(e2).__name__}: {e2}"[:200]

    sharded = None
    if world > 1 and args.gather != "none" and not args.no_extras:
        # C4 / C5 per rank (one RCCL communicator for C4's gather-to-root); every rank reports, rank 0 prints
        try:
            wd.arm(args.phase_timeout, "C4 sharded errstate gather / C5 per-rank share", rank)
            comm2 = P.Comm.from_torch_distributed(local_rank, stream=torch.cuda.current_stream(dev).cuda_stream)
            mine = sharded_extras(dev, rank, world, comm2, red_dev)
            comm2.close()
            box = [None] * world
            dist.all_gather_object(box, {"rank": rank, **{k: v.get("value") for k, v in mine.items()}})
            sharded = dict(mine); sharded["per_rank_values"] = box
        except Exception as e:
            sharded = {"error": f"{type(e).__name__}: {e}"[:300]}
    wd.arm(args.phase_timeout, "secondary metrics / CPU baseline", rank)

    extras = small = None
    if rank == 0 and not args.no_extras:
        try:
            # The headline's context (its own stream + the overlap mode's internal one) has done its work: close it before the secondary metrics.  HIP maps a
            # process's streams onto FOUR hardware queues; with those two still alive, the child streams of config 5's one-call form (one per model, one for
            # the pipelined pieces) landed on a queue that was already taken and its kernels ran one behind the other: 290 us in this process where the same
            # call takes 260 us in a process of its own (tools/c5_one_call.py, profiles/r05/c5.jsonl -- same box, same run).
            bctx.synchronize(); bctx.close()
            extras, small = secondary_metrics(dev, args.extras_reps)
        except Exception as e:
            extras = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        out = line
        if gather:
            out["gather"] = gather
        if sharded:
            out["sharded_configs"] = sharded
        if extras:
            out["extras"] = extras
        if small:
            out["small_batches_fused"] = small
        if overlapped:
            out["independent_batches_multi_stream"] = overlapped
        if not args.no_cpu_baseline and world == 1:
            wd.arm(max(args.phase_timeout, 120.0), "CPU baseline", rank)
            out["cpu_baseline"] = cpu_baseline(Zn, t, dt)
        wd.disarm()
        print(json.dumps(out), flush=True)
    wd.disarm()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
