#!/usr/bin/env python3
"""tools/body_probe.py -- body-frame rigid bodies, RK4 Jacobians at 65 536 knots: ForwardAD (duals through the step) against the UserDefined chain rule
(analytic stage Jacobians, matrix-free) and FiniteDifference, same box, one JSON line each.  Round 6: which body should ForwardAD run on where the dual
kernel spills?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import rdhip as RD  # noqa: E402
import bench_configs as BC  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(12)
# the clocks first (a 10-ms measurement inside the ramp after an idle period reads 5-10 % slow, and this tool's first rows used to): half a second of the flagship
import sys as _sys, io as _io
_out, _sys.stdout = _sys.stdout, _io.StringIO()
try:
    for _ in range(4):
        BC.jac_case("warm-up", RD.Satellite(), BC.rigid_inputs(N, 6, rng, 0.0, 1.0), RD.RK4, RD.ForwardAD(), np.float64, 1500)
finally:
    _sys.stdout = _out
trim = 9.81 * 0.5 / 4
for tag, model, m, lo, hi in (("body-frame quadrotor", RD.Quadrotor(bodyframe=True), 4, 0.5 * trim, 1.5 * trim), ("world-frame quadrotor", RD.Quadrotor(), 4, 0.5 * trim, 1.5 * trim),
                              ("body-frame satellite", RD.Satellite(bodyframe=True), 6, 0.0, 1.0), ("world-frame satellite", RD.Satellite(), 6, 0.0, 1.0)):
    Zn = BC.rigid_inputs(N, m, rng, lo, hi)
    for dname, diff in (("ForwardAD", RD.ForwardAD()), ("UserDefined", RD.UserDefined()), ("FiniteDifference", RD.FiniteDifference())):
        for rep in range(2):
            BC.jac_case(f"{tag} RK4 jacobian {dname}", model, Zn, RD.RK4, diff, np.float64, 1000)
