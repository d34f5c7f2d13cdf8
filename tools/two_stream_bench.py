#!/usr/bin/env python3
"""tools/two_stream_bench.py -- independent batches issued alternately on two HIP streams (one rdhip context per stream): the next
batch's workgroups start on the CUs the previous launch has already left, so its prologue runs under the previous launch's tail."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rdhip as RD  # noqa: E402
from tools.bench_configs import rigid_inputs  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(0)
sat = RD.Satellite()
dm = RD.DiscretizedDynamics(sat, RD.RK4)
Z = RD.PackedTrajectory(13, 6, rigid_inputs(N, 6, rng), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
NB = 4
Js = [RD.alloc_jacobians(dm, N, Z.data) for _ in range(NB)]
Ys = [torch.empty((N, 13), dtype=torch.float64, device="cuda:0") for _ in range(NB)]
sig, AD = RD.StaticReturn(), RD.ForwardAD()


def run(nstreams):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    calls = []
    for b in range(NB):
        with torch.cuda.stream(streams[b % nstreams]):
            calls.append(RD.prepare_jacobian(sig, AD, dm, Js[b], Ys[b], Z))
    for i in range(20):
        with torch.cuda.stream(streams[i % nstreams]):
            calls[i % NB]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        with torch.cuda.stream(streams[i % nstreams]):
            calls[i % NB]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for ns in (1, 2, 4, 1, 2, 4):
    us = run(ns)
    print(json.dumps({"streams": ns, "N": N, "us_per_batch": round(us, 2), "frac_of_8TBps": round(N * 2248 / (us * 1e-6) / 8e12, 3)}))
