#!/usr/bin/env python3
"""tools/pmc_valu.py -- a few launches each of the kernels whose second roofline is the fp64 vector pipe (run under rocprofv3 --pmc SQ_* by tools/pmc_valu.sh):
the flagship (world-frame satellite), the world- and body-frame quadrotor, the body-frame satellite, the ImplicitMidpoint + IFT kernel; 65 536 knots, fp64."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import numpy as np
import torch

import rdhip as RD
import bench_configs as BC

N, REPS = 65536, 6
rng = np.random.default_rng(21)
sig, AD = RD.StaticReturn(), RD.ForwardAD()
trim = 9.81 * 0.5 / 4
for tag, model, m, lo, hi, rule, diff in (
        ("satellite world RK4 ForwardAD", RD.Satellite(), 6, 0.0, 1.0, RD.RK4, AD),
        ("quadrotor world RK4 ForwardAD", RD.Quadrotor(), 4, 0.5 * trim, 1.5 * trim, RD.RK4, AD),
        ("quadrotor body RK4 ForwardAD", RD.Quadrotor(bodyframe=True), 4, 0.5 * trim, 1.5 * trim, RD.RK4, AD),
        ("satellite body RK4 ForwardAD", RD.Satellite(bodyframe=True), 6, 0.0, 1.0, RD.RK4, AD),
        ("satellite world ImplicitMidpoint IFT", RD.Satellite(), 6, 0.0, 1.0, RD.ImplicitMidpoint, RD.ImplicitFunctionTheorem(AD))):
    fun = RD.DiscretizedDynamics(model, rule)
    Z = RD.PackedTrajectory(13, m, BC.rigid_inputs(N, m, rng, lo, hi), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda")
    J = RD.alloc_jacobians(fun, N, Z.data); Y = torch.empty((N, 13), dtype=torch.float64, device="cuda")
    for _ in range(REPS):
        RD.jacobian_(sig, diff, fun, J, Y, Z)
    torch.cuda.synchronize()
    print(tag, "|", RD.last_kernel_name()[:160], flush=True)
